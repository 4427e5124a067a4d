/*
 * fdnn.h -- C-ABI of the MI355X-native fast-dnn scorer (libfast-dnn.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ / torch types.
 * The JNI exports the Java class suskun.nn.QuantizedDnn binds
 * (include/fdnn_jni.h) are thin wrappers over these functions, and each entry
 * point cites the reference interface it replaces (paths relative to the
 * reference repository).
 *
 * Conventions
 *   - every function returns FDNN_OK (0) or a negative fdnn_status; the text of
 *     the last error on the calling thread is in fdnn_last_error().
 *   - "host" pointers are ordinary process memory; "device" pointers (d_*) are
 *     HIP device memory on the model's device.  Inputs are const: unlike the
 *     reference (dnn.cc:175-192) the caller's frames are never modified.
 *   - frames are row-major n x input_dim fp32, input_dim being the padded
 *     (multiple of 4) layer-0 width the reference reports (jni_dnn.cc:20-25);
 *     outputs are row-major n x output_dim fp32 soft-max rows.
 *   - a model handle is immutable after load and may be used from many threads
 *     at once (jni_dnn.cc:49-51 gives every call its own context; here calls
 *     draw a device context from a per-model pool).  A context handle is
 *     single-threaded, like the reference's LazyContext.
 *   - stream ordering on a context: the *_device entry points enqueue on the
 *     caller's stream and do not synchronize; the host-pointer entry points use
 *     the context's own stream and return with their result complete.  Calls on
 *     ONE context are ordered behind each other whatever streams they use (each
 *     waits, on the device, for the context's previously enqueued work), so
 *     forward_hidden_device(stream) followed by lazy_output() / read_hidden()
 *     needs no synchronization by the caller.  Caller-owned buffers (d_x,
 *     d_masks, d_out) are the caller's to order.
 *   - there is no CPU fallback: without a usable HIP device every entry point
 *     that computes returns FDNN_E_DEVICE.
 */
#ifndef FDNN_H
#define FDNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDNN_API __attribute__((visibility("default")))

typedef enum fdnn_status {
  FDNN_OK = 0,
  FDNN_E_ARG = -1,    /* null / out-of-range argument, dimension mismatch */
  FDNN_E_IO = -2,     /* file cannot be opened / short read */
  FDNN_E_FORMAT = -3, /* .bin is not a net this path can run (see fdnn_model_load) */
  FDNN_E_DEVICE = -4, /* no HIP device, or a HIP call failed */
  FDNN_E_NOMEM = -5,
  FDNN_E_STATE = -6   /* call sequence violated (e.g. lazy output before forward_hidden) */
} fdnn_status;

typedef struct fdnn_model fdnn_model; /* replaces dnn::QuantizedDnn*   (dnn.h:106-142) */
typedef struct fdnn_ctx fdnn_ctx;     /* replaces dnn::CalculationContext* (dnn.h:144-208) */

FDNN_API const char *fdnn_last_error(void);
FDNN_API const char *fdnn_version(void);
FDNN_API int fdnn_device_count(void);

/* ------------------------------------------------------------------ model
 * fdnn_model_load <- Java initialize(String, float) / jni_dnn.cc:7-18:
 *   FloatDnn(path) (float_dnn.cc:18-69) + QuantizedDnn(floatDnn, cutoff)
 *   (dnn.cc:511-531).  Reads the big-endian .bin, pads layer-0 input to x4,
 *   quantizes layers 1.. with the reference's per-layer multiplier rule
 *   (dnn.cc:460-509) and uploads one packed weight blob to `device`.
 *   Accepts what the reference can run: >= 4 affine layers (dnn.cc:199 needs
 *   layers()[1] to be a hidden layer), all hidden widths equal and x16
 *   (README.md:10,:69); otherwise FDNN_E_FORMAT.  cutoff must be > 0
 *   (QuantizedDnn.java:55-57). */
FDNN_API int fdnn_model_load(const char *path, float cutoff, fdnn_model **out);
FDNN_API int fdnn_model_load_on(const char *path, float cutoff, int device, fdnn_model **out);
/* <- Java delete() / jni_dnn.cc:128-133.  Contexts must be freed first. */
FDNN_API void fdnn_model_free(fdnn_model *m);

/* <- inputDimension / outputDimension / layerCount / layerDimension,
 *    jni_dnn.cc:20-33, :135-156.  layer_count = quantized layers + 1;
 *    layer_dim(0) = layer-0 node count, layer_dim(k>=1) = node count of
 *    quantized layer index k as the reference indexes it (layers()[k], i.e.
 *    affine layer k+1); out of range -> -1.  The reference reads one past the
 *    end for k == layer_count-1 (its bound check is off by one); here that
 *    index returns -1. */
FDNN_API int fdnn_model_input_dim(const fdnn_model *m);
FDNN_API int fdnn_model_output_dim(const fdnn_model *m);
FDNN_API int fdnn_model_hidden_dim(const fdnn_model *m);
FDNN_API int fdnn_model_layer_count(const fdnn_model *m);
FDNN_API int fdnn_model_layer_dim(const fdnn_model *m, int index);
FDNN_API int fdnn_model_device(const fdnn_model *m);

/* Numeric flavour of the fp32 input layer.
 *   0 (default) -- unfused multiply then add, four k-mod-4 partial sums
 *                  combined (l0+l1)+(l2+l3): the reference built
 *                  -O2 -msse4 -ffp-contract=off (dnn.cc:219-247, :168-172).
 *   1           -- the same chains with fused multiply-add, i.e. the reference
 *                  built with its own Makefile's -march=native on an FMA host. */
FDNN_API int fdnn_model_set_l0_fma(fdnn_model *m, int on);

/* ------------------------------------------------------------------ dense path
 * fdnn_calculate <- Java calculate(long, float[], int, int, int) /
 *   jni_dnn.cc:35-62 -> CalculationContext::Calculate (dnn.cc:162-165).
 *   x: host n x dim, out: host n x output_dim.  dim must equal input_dim
 *   (QuantizedDnn.java:157-161).  n == 0 is a no-op.  batch_hint is the
 *   reference's frame-block size; results never depend on it. */
FDNN_API int fdnn_calculate(fdnn_model *m, const float *x, int n, int dim, int batch_hint, float *out);
/* Same computation on device-resident buffers (the roofline path): d_x and
 * d_out live on the model's device, work is enqueued on `stream` (a
 * hipStream_t; NULL = default stream) and NOT synchronized. */
FDNN_API int fdnn_calculate_device(fdnn_model *m, const float *d_x, int n, float *d_out, void *stream);

/* One-call lazy scoring: CalculateUntilLastHiddenLayer (dnn.cc:402-424) + LazyOutputActivations
 * (dnn.cc:355-392) for every frame of the call, in one call and one stream synchronisation -- what a LazyContext does in
 * two calls per utterance (QuantizedDnn.java:72-107), without the per-frame JNI round trips README.md:45 complains about.
 *   fdnn_calculate_lazy       masks [n][output_dim] bytes, non-zero = active (the JNI contract's byte masks)
 *   fdnn_calculate_lazy_bits  bits [n][ceil(output_dim / 64)] 64-bit words, bit b of word w = node 64 w + b
 * out [n][output_dim]: active nodes their probability, inactive nodes 1 / total (dnn.cc:366-369, :389).  Host callers get
 * their rows back COMPACTED over PCIe (active probabilities + one value per frame) and rebuilt in `out`.
 * The _device form works on device-resident buffers, enqueued on `stream`, not synchronised. */
FDNN_API int fdnn_calculate_lazy(fdnn_model *m, const float *x, int n, int dim, const int8_t *masks, float *out);
FDNN_API int fdnn_calculate_lazy_bits(fdnn_model *m, const float *x, int n, int dim, const uint64_t *bits, float *out);
FDNN_API int fdnn_calculate_lazy_bits_device(fdnn_model *m, const float *d_x, int n, const uint64_t *d_bits, float *d_out, void *stream);

/* ------------------------------------------------------------------ contexts / lazy path
 * fdnn_ctx_create <- getContext / jni_dnn.cc:64-77 (CalculationContext ctor,
 *   dnn.cc:194-215): device scratch for n frames. */
FDNN_API int fdnn_ctx_create(fdnn_model *m, int n, int batch_hint, fdnn_ctx **out);
/* <- deleteLazyContext / jni_dnn.cc:119-126 */
FDNN_API void fdnn_ctx_free(fdnn_ctx *c);
FDNN_API int fdnn_ctx_frame_count(const fdnn_ctx *c);
FDNN_API int fdnn_ctx_output_dim(const fdnn_ctx *c);
/* <- calculateUntilOutput / jni_dnn.cc:79-95 -> CalculateUntilLastHiddenLayer
 *    (dnn.cc:402-424).  x: host n x input_dim. */
FDNN_API int fdnn_ctx_forward_hidden(fdnn_ctx *c, const float *x);
FDNN_API int fdnn_ctx_forward_hidden_device(fdnn_ctx *c, const float *d_x, void *stream);
/* <- calculateLazy / jni_dnn.cc:97-117 -> LazyOutputActivations
 *    (dnn.cc:355-392): one frame, mask of output_dim bytes (non-zero = active).
 *    Masked-out nodes keep logit 0, contribute exp(0) to the soft-max
 *    denominator and come back as 1/total.  out: host output_dim floats.
 *    frame >= n leaves the reference with a partially written buffer; here it
 *    is FDNN_E_ARG. */
FDNN_API int fdnn_ctx_lazy_output(fdnn_ctx *c, int frame, const int8_t *mask, float *out);
/* Batched form of the same contract: frames
 * [first, first+count) with masks[count][output_dim] -> out[count][output_dim]. */
FDNN_API int fdnn_ctx_lazy_output_batch(fdnn_ctx *c, int first, int count, const int8_t *masks, float *out);
FDNN_API int fdnn_ctx_lazy_output_batch_device(fdnn_ctx *c, int first, int count, const int8_t *d_masks, float *d_out,
                                               void *stream);
/* The same with the masks as BITS: bits[count][ceil(output_dim / 64)] 64-bit words, bit b of word w of a row = node
 * 64 w + b active (bits past output_dim ignored).  A decoder that keeps its active set as a bit set hands it over as it
 * is: an eighth of the bytes over PCIe / HBM and no pack pass (LazyOutputActivations, dnn.cc:355-392: inactive nodes
 * contribute exp(0) to the sum and all read 1 / total). */
FDNN_API int fdnn_ctx_lazy_output_batch_bits(fdnn_ctx *c, int first, int count, const uint64_t *bits, float *out);
FDNN_API int fdnn_ctx_lazy_output_batch_bits_device(fdnn_ctx *c, int first, int count, const uint64_t *d_bits, float *d_out, void *stream);
/* Dense output layer over the context's hidden activations
 * (CalculateOutput, dnn.cc:428-454). */
FDNN_API int fdnn_ctx_output(fdnn_ctx *c, float *out);
FDNN_API int fdnn_ctx_output_device(fdnn_ctx *c, float *d_out, void *stream);
/* Last hidden layer's u8 activations, n x hidden_dim (quantized_activations_). */
FDNN_API int fdnn_ctx_read_hidden(fdnn_ctx *c, uint8_t *out);

/* ------------------------------------------------------------------ multi-stream scoring loop
 * Generalises the reference's serving shape -- caller threads over independent
 * utterances, one context per call (QuantizedDnn.java:72-107,
 * MultiThreadedStressTest.java:48-69) -- to batches in flight on one GPU.
 * A server owns `depth` slots (a context sized for max_frames each) and two
 * shared streams: consecutive large batches are serialised on the compute
 * stream while the HBM-bound soft-max scale of batch i runs on the tail stream
 * under the VALU-bound layer 0 of batch i+1; batches too small to fill the chip
 * run whole on their slot's stream, next to each other.  A submission returns a
 * ticket; fdnn_server_wait(ticket) returns when its results are complete.
 * At most `depth` device submissions are in flight: a further submit first
 * waits for the oldest.  All entry points are thread-safe.
 *
 *   fdnn_server_submit_device  d_x [n][input_dim] and d_out [n][output_dim] live
 *     on the model's device and must stay valid until the ticket completes;
 *     d_masks (may be NULL) [n][output_dim] selects the lazy contract
 *     (dnn.cc:355-392).  n <= max_frames.
 *   fdnn_server_submit         host buffers, any n >= 1.  Submissions from any
 *     number of threads are coalesced: packed into one batch of up to
 *     max_frames frames per slot, scored once, scattered back to each caller's
 *     `out`.  x / masks / out must stay valid until the ticket completes.  A
 *     coalesced utterance is bit-identical to the same utterance scored alone
 *     (frames are independent, kernels are batch-size invariant).
 *   fdnn_server_set_linger_us  how long the packer waits for more host
 *     submissions before launching a batch that is not full (default 0); it
 *     never lingers while the server is idle (no batch in flight).
 *   fdnn_model_enable_batcher  routes fdnn_calculate (= the JNI calculate()) of
 *     this model through an internal server, so that the unmodified Java class
 *     called from many threads is coalesced as well; also switched on at load by
 *     the environment variable FDNN_BATCHER=max_frames[:depth[:linger_us]]. */
typedef struct fdnn_server fdnn_server;
FDNN_API int fdnn_server_create(fdnn_model *m, int max_frames, int depth, fdnn_server **out);
FDNN_API void fdnn_server_free(fdnn_server *s);
FDNN_API int fdnn_server_set_linger_us(fdnn_server *s, int microseconds);
FDNN_API int fdnn_server_submit_device(fdnn_server *s, const float *d_x, int n, const int8_t *d_masks, float *d_out,
                                       uint64_t *ticket);
FDNN_API int fdnn_server_submit(fdnn_server *s, const float *x, int n, const int8_t *masks, float *out, uint64_t *ticket);
/* The lazy contract through the loop with the masks as BITS (bits [n][ceil(output_dim / 64)], bit b of word w = node
 * 64 w + b, as fdnn_calculate_lazy_bits): bit-mask submissions are coalesced with one another, scored once, and come back
 * COMPACTED -- the active nodes' probabilities and one value per frame over PCIe, each caller's rows rebuilt inside its
 * own `out` by the thread that waits for the ticket.  x / bits / out must stay valid until the ticket completes.
 * (LazyContext.calculateUntilOutput + calculateForOutputNodes per frame, QuantizedDnn.java:72-107, for many callers.) */
FDNN_API int fdnn_server_submit_lazy_bits(fdnn_server *s, const float *x, int n, const uint64_t *bits, float *out, uint64_t *ticket);
FDNN_API int fdnn_server_wait(fdnn_server *s, uint64_t ticket);
FDNN_API int fdnn_server_drain(fdnn_server *s);
FDNN_API int fdnn_server_stats(fdnn_server *s, uint64_t *batches, uint64_t *frames, uint64_t *requests,
                               uint64_t *coalesced_requests);
FDNN_API int fdnn_model_enable_batcher(fdnn_model *m, int max_frames, int depth, int linger_us);

/* ------------------------------------------------------------------ one process, N devices of one node
 * BASELINE north_star: frames shard across the node's GPUs, weights broadcast at
 * load only.  fdnn_group_load quantizes on devices[0] and sends the packed blob
 * device-to-device to every other listed device (hipMemcpyPeer over xGMI, or one
 * ncclBroadcast when FDNN_GROUP_BCAST=rccl); fdnn_group_calculate cuts the n
 * frames of ONE call (host buffers, as fdnn_calculate / jni_dnn.cc:35-62) into
 * contiguous shards -- fdnn_group_shard, sizes differ by at most one -- scores
 * them concurrently, one host thread per device, each into its slice of `out`.
 * No collective in steady state.  A device may be listed more than once (two
 * replicas on one GPU: how the protocol is tested on a one-GPU box).
 * fdnn_group_attach makes the group transparent: fdnn_calculate on
 * fdnn_group_model(g, 0) shards over the group and fdnn_model_free on it frees
 * the whole group -- which is what fdnn_model_load does by itself when the
 * environment variable FDNN_DEVICES lists several devices ("0,1,2,3" / "all"),
 * so the unmodified Java class scales with no code change. */
typedef struct fdnn_group fdnn_group;
FDNN_API int fdnn_group_load(const char *path, float cutoff, const int *devices, int n_devices, fdnn_group **out);
FDNN_API void fdnn_group_free(fdnn_group *g);
FDNN_API int fdnn_group_size(const fdnn_group *g);
FDNN_API fdnn_model *fdnn_group_model(const fdnn_group *g, int index);
FDNN_API const char *fdnn_group_weight_transport(const fdnn_group *g); /* "none" | "peer-copy" | "rccl" */
/* CPU list the persistent host thread of replica `index` pinned itself to (its device's NUMA-local CPUs), "" before the
 * first large call or when the list could not be read.  Diagnostics. */
FDNN_API const char *fdnn_group_worker_cpus(fdnn_group *g, int index);
FDNN_API int fdnn_group_calculate(fdnn_group *g, const float *x, int n, int dim, int batch_hint, float *out);
FDNN_API void fdnn_group_shard(int n, int world, int rank, int *start, int *stop);
FDNN_API int fdnn_group_attach(fdnn_group *g);

/* ------------------------------------------------------------------ multi-GPU weight distribution
 * Rank 0 quantizes once; the packed blob (header + fp32 layer 0 + int8 layers +
 * per-node offsets + biases + shift/scale + LUT) is broadcast over RCCL by the
 * caller and imported on every other device, so all ranks hold bit-identical
 * weights.  No reference counterpart (the reference is single-process). */
FDNN_API int fdnn_model_blob_size(const fdnn_model *m, size_t *bytes);
FDNN_API int fdnn_model_export_blob(const fdnn_model *m, void *d_dst, size_t capacity, void *stream);
FDNN_API int fdnn_model_import_blob(const void *d_src, size_t bytes, int device, fdnn_model **out);

/* ------------------------------------------------------------------ parity taps (same kernels, extra stores)
 * Runs the dense path on host buffers and also returns intermediate state;
 * any output pointer may be NULL.
 *   l0_lin   [n][H]              layer-0 activation after bias (fp32)
 *   u8_acts  [n_hidden][n][H]    u8 activations after every hidden layer
 *   acc_hid  [n_hidden-1][n][H]  int32 accumulators of the int8 hidden layers
 *                                (pmaddubsw-saturating semantics, dnn.cc:323-349)
 *   acc_out  [n][O]              same for the output layer
 *   logits   [n][O]              output value after bias, before soft-max
 *   probs    [n][O]
 * masks (may be NULL) switches the output layer to the lazy contract. */
FDNN_API int fdnn_debug_forward_taps(fdnn_model *m, const float *x, int n, const int8_t *masks, float *l0_lin,
                                     uint8_t *u8_acts, int32_t *acc_hid, int32_t *acc_out, float *logits, float *probs);

/* The int32 accumulators of the output layer as the PRODUCTION kernel instances hold them (no tap kernels anywhere on
 * the path: the dense / masked instance a plain call of the same size launches), for every stride-th frame:
 * acc [ceil(n/stride)][O].  probs (may be NULL) receives the call's ordinary result [n][O].  Parity tests only. */
FDNN_API int fdnn_debug_production_acc_out(fdnn_model *m, const float *x, int n, int stride, const int8_t *masks, int32_t *acc,
                                           float *probs);

/* How a pass over n frames is cut into chunks (host logic only, no device needed): chunks[2 i] = first frame,
 * chunks[2 i + 1] = frame count of chunk i; returns the number of chunks, or -1 if `cap` pairs do not hold them.  The
 * reference blocks a call by `batch` frames (dnn.cc:402-454: blocking never changes a result); here a very large batch
 * runs as device-sized chunks for cache locality, results unchanged (tests / diagnostics). */
FDNN_API int fdnn_debug_frame_chunks(int n, int *chunks, int cap);
/* The same for a batch whose hidden layers run (chained != 0) or do not run as one chained launch: without the chain a few
 * frames past a whole round of workgroups (up to 2 048) go as a batch of their own (host logic; fdnn_debug_set_chain(0),
 * FDNN_CHAIN=0, fewer than two int8 hidden layers, a layer without the validated division are such configurations). */
FDNN_API int fdnn_debug_frame_chunks_for(int n, int chained, int *chunks, int cap);

/* Which kernel computes the canonical fp32 input layer (tests / measurements only; results are
 * bit-identical): 0 = by batch size (default), 1 = always the chain-pass kernel (128 x 128
 * tiles over chain-major images), 2 = always the 64 x 64-tile kernel, 3 = the screened matrix-pipe path wherever it is
 * available (batches of 2048 frames and more, no taps). */
FDNN_API int fdnn_debug_set_l0_kernel(fdnn_model *m, int kind);

/* Tests only, process-wide: 0 = large batches scale their soft-max in a separate pass (what FDNN_FUSE_NORM=0 or a second
 * process on the GPU selects), 1 = inside the output kernel wherever the shape allows, -1 = the default rule.  Results are
 * bit-identical (one tree order for the row total everywhere).  SoftMax::apply, dnn.cc:534-544. */
FDNN_API int fdnn_debug_set_fuse(int mode);

/* Tests only, host code, no device needed: the host half of a compacted lazy return (what fdnn_calculate_lazy_bits and
 * the scoring loop run on the caller's thread).  comp [count][stride] = per row its inactive value, then its active nodes'
 * values in node order; bits [count][ceil(O / 64)]; out [count][O].  mode 0: the library's choice (AVX-512 expanding loads
 * where the CPU has them), 1: the scalar form, 2: comp copied into the tail of out first (the in-place form of the
 * one-call entry).  LazyOutputActivations' row layout, dnn.cc:366-369, :389. */
FDNN_API int fdnn_debug_lazy_expand(float *out, const float *comp, int count, int O, int stride, const uint64_t *bits, int mode);

/* Tests only: cap the per-launch list of flagged layer-0 outputs (int8 screening) of contexts created AFTER the call at
 * `cap` entries (0 = the default, 1/16 of the outputs), so that a small batch overflows it: tiles that no longer fit take
 * the whole-tile recomputation, results unchanged.  Drops the model's pooled contexts.  InputActivations, dnn.cc:219-247. */
FDNN_API int fdnn_debug_set_l0_list_cap(fdnn_model *m, int cap);

/* How the int8 hidden layers run (tests / measurements only; results are bit-identical): mode 1 = as ONE persistent
 * launch (fdnn_chain.hip: tasks drawn from per-XCD queues, a task waits only for its own frame tile's node tiles of the
 * layer before) for batches of at least min_frames frames (<= 0: the default threshold), 0 = one launch per layer,
 * -1 = the default (FDNN_CHAIN / FDNN_CHAIN_MIN in the environment, else on).  Process-wide.
 * CalculateUntilLastHiddenLayer's layer loop, src/cpp/dnn.cc:413-423, is what is being computed. */
FDNN_API int fdnn_debug_set_chain(int mode, int min_frames);

/* How a large batch's int8 hidden layers run when they are launched layer by layer: 1 = the role-split kernel (fdnn_pp.hip:
 * one wave of each SIMD in the k-loop, its partner staging that tile's operands and running the previous tile's epilogue)
 * for batches of at least min_frames frames (<= 0: the default threshold), 0 = fdnn_gemm.hip's in-phase tiles, -1 = the
 * default (FDNN_PP / FDNN_PP_MIN in the environment, else: layers without saturating pairs from 16 384 frames).  Process-wide; identical bytes either way.
 * QuantizedLayerActivations + AddBias + QuantizedSigmoid, src/cpp/dnn.cc:250-349, is what is being computed. */
FDNN_API int fdnn_debug_set_pp(int mode, int min_frames);

/* How the OUTPUT layer of a large dense batch runs when its soft-max is fused: 1 = the role-split kernel (fdnn_ppo.hip: the
 * exchange of the 256-node row sums and the scale run beside the next half's k-loop) whenever the shape allows (8 000-node
 * class layer: 32 node tiles, 2 048 inputs, dense, validated division), 0 = fdnn_gemm.hip's in-phase fused tiles, -1 = the
 * default (FDNN_PPO in the environment, else by batch size: a layer without saturating weight pairs from 14 pairs of 160-frame
 * halves = 4 161 frames, one with pairs from 22 = 6 721, in either case only where the launch's last round of frame pairs is
 * at least 3/4 resp. 4/5 full -- fdnn_ppo.hip: qppo_ok).  Process-wide; identical bits either way.
 * CalculateOutput + SoftMax::apply, src/cpp/dnn.cc:428-454, :534-544, is what is being computed. */
FDNN_API int fdnn_debug_set_ppo(int mode);

/* Tests: write the word a fused soft-max workgroup raises (in host memory) when it gives up waiting for its frame tile's
 * siblings -- 1: as if a launch of this model had just done so (from the next call on the model runs the unfused soft-max
 * and says so once on stderr), 0: forget it.  The production path needs no call: fdnn_gemm.hip / fdnn_ppo.hip raise it. */
FDNN_API int fdnn_debug_raise_fuse_fault(fdnn_model *m, int value);

/* Measurement builds of the chained hidden-layer kernel (-DFDNN_CHAIN_CLK=1; the shipped library records nothing): with
 * out == NULL, give the context a buffer for the phase clocks of cap_tasks tasks; with out != NULL copy the records of the
 * launches since ([0] = tasks recorded, then from [8] ten words per task: block / task id, XCD / layer / frame tile, the
 * wall clock, seven cycle stamps) into out[8 + 10 * cap_tasks] and start over. */
FDNN_API int fdnn_debug_chain_clocks(fdnn_ctx *c, long long *out, int cap_tasks);

/* Layer 0 alone through the PRODUCTION kernels (no taps): u8_out [n][hidden_dim].  Large batches take the
 * screened path (fused chains on the fp32 matrix pipe, a rigorous bound on |fused - unfused|, exact unfused
 * recomputation of the outputs whose table index the difference could change); *recomputed (may be NULL)
 * receives how many outputs of this call were recomputed.  Tests compare u8_out with the oracle bit for bit. */
FDNN_API int fdnn_debug_layer0(fdnn_model *m, const float *x, int n, uint8_t *u8_out, unsigned long long *recomputed);

/* The int8 screening of the input layer, opened up for the parity tests: besides u8_out [n][hidden_dim] (as
 * fdnn_debug_layer0, int8-screened path at any batch size) t_out / dd_out [n][hidden_dim] receive what the screening kernel
 * computed per output -- t~ = 100 lin~ from the exact digit products, and Dd, the half-width its error bound vouches for
 * (fdnn_l0s.hip).  The bound's claim is |100 lin_ref - t~| <= Dd for the reference's lin (dnn.cc:219-264); the tests assert
 * it on every output and report the slack max |100 lin_ref - t~| / Dd.  Needs an input width of 64 .. 496. */
FDNN_API int fdnn_debug_layer0_screen(fdnn_model *m, const float *x, int n, uint8_t *u8_out, float *t_out, float *dd_out,
                                      unsigned long long *recomputed);

/* 1 when ANOTHER process held this GPU's marker (/dev/shm/fdnn-gpu-<pci bus id>, an advisory lock taken by the first process
 * that asks / loads a model there) when this process first looked: this process then scales its soft-max in a separate pass
 * instead of inside the output kernel, whose workgroups wait for one another and must not share the chip with another
 * process's (same results).  0 = this process owns the marker, or none could be created.
 * Negative = error.  The reference's concurrency model is threads of one process (MultiThreadedStressTest.java:48-69). */
FDNN_API int fdnn_device_shared(int device);

/* Fused soft-max health counter.  Large dense / batched-lazy calls scale their soft-max inside the output kernel: the
 * node tiles of a frame tile exchange row sums and wait for one another (bounded).  Within a process the library chains
 * those launches per device, so the wait is microseconds; a workgroup whose wait nevertheless timed out (another PROCESS
 * on the same GPU, see INTEGRATION.md: FDNN_FUSE_NORM=0) leaves its block to the frame tile's last workgroup -- same bits, tens of
 * milliseconds late.  *tiles = how many tiles that has happened to on this model since load (0 in a healthy setup).
 * SoftMax::apply, src/cpp/dnn.cc:534-544, is what is being computed. */
FDNN_API int fdnn_model_fuse_giveups(fdnn_model *m, unsigned long long *tiles);
/* Chained hidden layers health counter (fdnn_chain.hip: the int8 hidden layers of a large batch in one persistent launch,
 * a task waiting -- bounded, seconds -- only for its own frame tile's node tiles of the layer before).  *faults = waits that
 * ran into their bound on this model since load: 0 in a healthy setup; non-zero means a launch computed on rows that may
 * not have been written (counters left dirty by a killed launch, a wedged device).  The context that saw it re-zeroes its
 * counters and stops chaining; fdnn_calculate re-runs the affected pass before it returns; callers of the *_device entry
 * points, which do not synchronise, read this after their own synchronisation.
 * CalculateUntilLastHiddenLayer's layer loop, src/cpp/dnn.cc:413-423, is what is being computed. */
FDNN_API int fdnn_model_chain_faults(fdnn_model *m, unsigned long long *faults);
/* The model's raw device counter words, n <= 32 ([1] layer-0 outputs recomputed, [2] fused soft-max give-ups; kernel clock
 * stamps of timing builds from [4]).  Measurements only. */
FDNN_API int fdnn_debug_device_counters(fdnn_model *m, unsigned long long *out, int n);

/* ------------------------------------------------------------------ per-kernel timing (bench / profiling only)
 * Between begin and end every kernel launch of this model is bracketed by HIP
 * events recorded on the stream it is launched on; end() synchronizes them and
 * returns the summed device time and launch count per kernel kind. */
enum { FDNN_PROF_L0 = 0, FDNN_PROF_FIX = 1, FDNN_PROF_HIDDEN = 2, FDNN_PROF_OUTPUT = 3, FDNN_PROF_NORMALIZE = 4, FDNN_PROF_KINDS = 5 };
FDNN_API int fdnn_profile_begin(fdnn_model *m);
FDNN_API int fdnn_profile_end(fdnn_model *m, double *ms /*[FDNN_PROF_KINDS]*/, int *launches /*[FDNN_PROF_KINDS]*/);

/* ------------------------------------------------------------------ host-only helpers (no device needed)
 * The load-time half of the path, exposed so it can be checked on a machine
 * without a GPU: .bin parsing, the quantizer, the sigmoid table, and the
 * packed-blob sections the kernels read. */
typedef struct fdnn_host_model fdnn_host_model;
FDNN_API int fdnn_host_model_load(const char *path, float cutoff, fdnn_host_model **out);
FDNN_API void fdnn_host_model_free(fdnn_host_model *hm);
FDNN_API int fdnn_host_model_layers(const fdnn_host_model *hm); /* affine layers incl. fp32 layer 0 */
FDNN_API int fdnn_host_model_layer_in(const fdnn_host_model *hm, int j);
FDNN_API int fdnn_host_model_layer_out(const fdnn_host_model *hm, int j);
FDNN_API float fdnn_host_model_multiplier(const fdnn_host_model *hm, int j);            /* j >= 1 */
FDNN_API int fdnn_host_model_weights_q(const fdnn_host_model *hm, int j, int8_t *out);  /* out_dim x in_dim */
FDNN_API int fdnn_host_model_bias(const fdnn_host_model *hm, int j, float *out);
FDNN_API int fdnn_host_model_wsum128(const fdnn_host_model *hm, int j, int32_t *out);   /* 128 * sum_k w[node][k] */
FDNN_API long long fdnn_host_model_risky_pairs(const fdnn_host_model *hm, int j);       /* pairs that can saturate */
FDNN_API size_t fdnn_host_model_blob_size(const fdnn_host_model *hm);
/* The packed weight blob as bytes (what fdnn_model_export_blob hands to RCCL),
 * and the receiver-side validation of such bytes (magic/version/size), with no
 * device involved -- used by the world_size-2 gloo tests of the load protocol. */
FDNN_API int fdnn_host_model_blob(const fdnn_host_model *hm, void *out, size_t capacity);
FDNN_API int fdnn_host_blob_check(const void *bytes, size_t size, int *input_dim, int *hidden_dim, int *output_dim,
                                  int *n_affine);
FDNN_API int fdnn_host_sigmoid_lut(uint8_t *out1280); /* QuantizedSigmoid table, dnn.cc:100-115 */
FDNN_API int fdnn_host_quantize(const float *w, int rows, int cols, float cutoff, int8_t *out, float *multiplier);

#ifdef __cplusplus
}
#endif
#endif /* FDNN_H */
