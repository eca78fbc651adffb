/*
 * dv3hip.h -- C ABI of the MI355X (gfx950) DeepVoice3 hot-path library (libdv3hip.so).
 *
 * The reference (r9y9/deepvoice3_pytorch) has no FFI: its hot path is stock torch ops
 * called from Python modules.  This header is the boundary a maintainer would bind
 * (ctypes) to replace those torch calls; every entry point cites
 * the reference lines it stands in for (paths relative to the reference repo root).
 *
 * Conventions
 *  - plain C: device pointers + sizes only, no torch types.  All tensors are fp32
 *    unless a name says otherwise; "BCT" = (batch, channel, time), time contiguous --
 *    the layout the reference conv stacks use (deepvoice3_pytorch/modules.py:139-164).
 *  - every call enqueues on `stream` (a hipStream_t passed as void*), never syncs,
 *    never allocates, never frees, retains no pointer after returning.
 *  - return value: 0 on success, a negative DV3_E* code otherwise;
 *    dv3_last_error() returns a thread-local message for the last failure.
 *  - strides are in ELEMENTS.  "bs" = batch stride, "rs" = row (channel) stride.
 */
#ifndef DV3HIP_H
#define DV3HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DV3_OK 0
#define DV3_EINVAL (-1)   /* bad argument / unsupported shape */
#define DV3_ELAUNCH (-2)  /* hipLaunch / runtime error        */

/* ABI version, bumped on any struct change; checked by the Python loader. */
#define DV3_ABI_VERSION 43
int dv3_abi_version(void);
const char* dv3_last_error(void);
/* Fills name (<=255 chars) of device `dev`, number of CUs; returns 0/err. */
int dv3_device_info(int dev, char* name, int name_len, int* n_cu);
/* sizeof(struct <name>) or -1: lets a foreign-language mirror of the descriptors self-check */
int dv3_sizeof(const char* name);
/* Developer knobs for measurements (what = 1: ablation variant of the bf16x3 tap-GEMM, 0 = off;
 * what = 2: bf16x3 wgrad tile, 0 auto / 1 = 128x128 / 2 = 256x128; what = 3: 8-wave bf16x3
 * tap-GEMM tiles on the in-phase (0) or ping-pong (1, default) main loop; what = 4: tile of the planes
 * tap-GEMM, 0 auto / 1 = 128x128 (4 waves, two workgroups per CU) / 2 = 128x64 / 9 = 128x256 (8 waves);
 * what = 5: start-up stagger of the second co-resident workgroup, -1 auto / n = n sleeps of ~4 us;
 * what = 34: conv_c8pp form, 0 = 8 waves on 256x256 only / 1 = two 4-wave workgroups per CU on 256x128 / 2 = by rule;
 * what = 40: launch census of dv3_conv_gemm_f32, 1 = clear and record / 0 = stop; what = 42: relative cost (percent)
 * of the 256x128 ping-pong tile in the split kernels' tile picker; what = 44: k-split form of the 128x64 split tile,
 * 0 never / 1 by rule (default) / 2 wherever eligible.  The full list: INTEGRATION.md, section 2). */
int dv3_debug_set(int what, int value);
/* what = 1: phase timestamps left by the last dv3_debug_set(1, 10) launch of the 128x256 bf16x3 tile
 * ([8 waves][192 slots][2] uint64, host pointer).  what = 40 / 41: the launch census -- the recorded dv3_conv_desc
 * structs (bytes <= count * sizeof(dv3_conv_desc)) / the kernel variant (int32 each, encoded as dv3_debug_get(10))
 * that served each; count = dv3_debug_get(40). */
int dv3_debug_read(int what, void* dst, int64_t bytes);
/* what = 10: which kernel the LAST dv3_conv_gemm_f32 call of this process launched, encoded
 * family * 1000 + tile_id * 10 + pingpong; family 1 = exact-fp32 streaming kernel, 2 = exact-fp32
 * LDS-staged kernel, 3 = split-bf16 (3 MFMAs per product), 4 = bf16 (1 MFMA per product); tile ids as
 * dv3_conv_desc.tile_hint (9 = the 8-wave 128x256 tile); + 2 on the 128x64 split tile = its k-split form.  what = 11: same for dv3_wgrad_gemm_f32
 * (family 1 = exact fp32, 3 = split-bf16, 4 = bf16; tile 1 = 128x128, 2 = 256x128).  Tests use it to
 * assert that a shape was served by the kernel the benchmark times.  Returns the value (>= 0). */
int dv3_debug_get(int what);

/* Measurement stand-in for an n-rank RCCL ring all-reduce on ONE GPU (csrc/standin.hip; dist.RingStandin): `channels`
 * persistent workgroups of `threads` (256 | 512) threads copy `bytes` (= 2 (n-1)/n x the bucket) from `src` (read only,
 * wrapped) into `scratch` (wrapped) at `bytes_per_us` for all workgroups together (the links' pace, kept by spinning on
 * the 100 MHz wall clock).  Reduces nothing.  Not on the product path.                                            */
int dv3_ring_standin(const void* src, int64_t src_bytes, void* scratch, int64_t scratch_bytes, int64_t bytes,
                     int32_t channels, int32_t threads, float bytes_per_us, void* stream);

/* ------------------------------------------------------------------------------------
 * Epilogue modes of the tap-GEMM (dv3_conv_gemm_f32).
 * ------------------------------------------------------------------------------------ */
enum {
  DV3_EPI_LINEAR = 0,  /* y = acc + bias                      1x1 Conv1d / Linear          */
  DV3_EPI_RELU = 1,    /* y = relu(acc + bias)                Conv1d + nn.ReLU             */
  DV3_EPI_SIGMOID = 2, /* y = sigmoid(acc + bias)             last 1x1 + torch.sigmoid     */
  DV3_EPI_GLU = 3,     /* Conv1dGLU:  modules.py:157-164                                   */
  DV3_EPI_HIGHWAY = 4, /* HighwayConv1d: modules.py:224-226                                */
  DV3_EPI_DGRAD = 5,   /* y = acc * dropmask(y-site) + addend   (input-gradient pass)      */
  DV3_EPI_SOFTSIGN = 6 /* y = v/(1+|v|), v = acc + bias       F.softsign(Linear(spk))      */
};
/* LINEAR/RELU/SIGMOID/SOFTSIGN: after the activation, if r  != NULL: y = (y + r ) * sqrt(.5)
 *                                               then if r2 != NULL: y = (y + r2) * sqrt(.5)
 * (AttentionLayer's `(x + residual) * sqrt(0.5)`, deepvoice3.py:175, and the decoder's outer
 *  residual, deepvoice3.py:348-349).                                                      */

enum {
  DV3_STORE_BCT = 0,       /* y[b][m][n]                                                   */
  DV3_STORE_INTERLEAVE2 = 1 /* y[b][m % Mo][2n + m / Mo], Mo = M/2: ConvTranspose1d k2 s2  */
};

/*
 * dv3_conv_gemm_f32 -- the hot kernel.  im2col-free dilated 1-D convolution as a tap-GEMM
 * on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), with the whole Conv1dGLU / HighwayConv1d
 * tail fused in the epilogue.
 *
 *   acc[b][m][n] = sum_{j<J} sum_{c<Cin} A[b][j][c][m] * xd[b][c][n + j*dil - padL]
 *   xd = x * bit(xmask) * drop_scale      (bit==1 everywhere when xmask == NULL)
 *
 * Replaces, per mode (reference file:line):
 *   GLU      F.dropout -> conv -> trim -> split -> (+softsign speaker bias) -> a*sigmoid(b)
 *            -> (x+residual)*sqrt(.5)                   deepvoice3_pytorch/modules.py:145-164
 *   HIGHWAY  same conv, T=sigmoid(b); T*a+(1-T)*x       deepvoice3_pytorch/modules.py:205-226
 *   LINEAR/RELU/SIGMOID  1x1 convs and Linear layers    deepvoice3.py:51-54,65-67,227-229,
 *            262-264,360-363,517,565-567,578,604; nyanko.py:29-31,96-100,133,149-156,365-398
 *   INTERLEAVE2 store     nn.ConvTranspose1d(k=2,s=2)   deepvoice3.py:519-520,527-528
 *   per-batch A (a_bs!=0) torch.bmm(q, keys)            deepvoice3.py:143
 *   DGRAD    input gradient of all of the above (autograd of F.conv1d + F.dropout)
 *
 * A is the PACKED weight: [J][Cin][lda], m contiguous, produced by dv3_weight_norm_pack_f32.
 * For GLU/HIGHWAY the m axis holds the `a` half in [0,Cg) and the gate half at [a_half,
 * a_half+Cg); M must equal 2*Cg and the output has Cg channels.
 */
typedef struct dv3_conv_desc {
  const float* x;  int64_t x_bs, x_rs;      /* input  [B][Cin][Tin]                         */
  const float* a;  int64_t a_bs; int32_t lda; int32_t a_half; /* packed weights            */
  const float* bias;                         /* [M] (reference order) or NULL               */
  const float* spk; int64_t spk_bs, spk_rs, spk_ts; /* additive on the `a` half (GLU) or NULL */
  const float* r;  int64_t r_bs, r_rs;       /* residual / highway input / DGRAD addend     */
  const float* r2; int64_t r2_bs, r2_rs;     /* second residual (non-gated modes) or NULL   */
  float* y;        int64_t y_bs, y_rs;       /* output                                      */
  float* ab;                                 /* optional save of pre-gate (a,b): [B][M][Tout] */
  const uint32_t* xmask; int32_t xmask_rs;   /* dropout keep-bits over x rows [B*Cin][rs]   */
  const uint32_t* ymask; int32_t ymask_rs;   /* DGRAD: keep-bits over y rows [B*M][rs]      */
  float drop_scale;                          /* 1/(1-p)                                     */
  int32_t B, Cin, Tin, M, Cg, Tout, J, dil, padL;
  int32_t mode, residual, store_mode;
  int32_t tile_hint;                         /* 0 = auto; else forces a tile config (tests) */
  const uint16_t* a_split;                   /* split-bf16 image of `a` (dv3_split_pack_bf16) or
                                                NULL.  Non-NULL selects the bf16x3 kernel (below) */
  int32_t split_terms;                       /* 0 or 3: hi/lo bf16 split, three MFMAs per product;
                                                1: hi planes only = plain bf16 MFMA with fp32 accumulate
                                                (BASELINE.json bf16 configs); DV3_SPLIT_F16X3 (19): `a_split`
                                                is a SCALED FP16 hi/lo image (below), the activations are
                                                split the same way while staging: three fp16 MFMAs per
                                                product, 2^-22-class operands = fp32-class results      */
  const uint16_t* x_planes;                  /* the input ALREADY split into operand planes (dv3_split_planes_f32
                                                layout, dtype matching split_terms; dropout already applied) or
                                                NULL.  Non-NULL (with a_split) selects the persistent planes
                                                kernel: both operands are staged with plain 16-byte copies.
                                                `x` may then be NULL unless the epilogue reads it as `r`.     */
  int32_t x_c8p;                             /* 8-channel blocks per batch item in x_planes (= round_up(Cin,32)/8) */
  float r_scale;                             /* DGRAD: y = acc * dropmask + r_scale * r (0 means 1).  The gradient
                                                that reaches a residual Conv1dGLU's input through the skip path is
                                                sqrt(.5) * dy: the epilogue reads dy itself instead of a scaled copy */
  int32_t io_bf16;                           /* bf16 STORAGE (BASELINE configs 3/4: "bf16 activations ... fp32 accum"),
                                                single-term bf16 kernels (split_terms == 1) only: DV3_IO_IN_BF16 = x, r
                                                and r2 are bf16 tensors (same element strides), DV3_IO_OUT_BF16 = y and
                                                ab are written as bf16 (round to nearest even).  0 = fp32 everywhere.  */
  const uint8_t* xmask_c8;                   /* c8 input (x_planes, split_terms == 1): dropout keep-BYTES [B][x_c8p][Tin],
                                                bit e of byte (b, g, t) = keep channel 8g+e at frame t
                                                (dv3_mask_bits_to_c8); applied while staging, 1/(1-p) = drop_scale in
                                                the epilogue.  NULL = no dropout.                                     */
  const uint8_t* ymask_c8;                   /* DGRAD with DV3_IO_OUT_C8: keep-bytes over the output rows            */
  void* sk_ws;                               /* optional stream-K workspace (dv3_conv_streamk_ws_bytes() bytes, its first
                                                4 KiB zero before the first launch that uses it; the launches leave them
                                                zero).  With it the 256 x 256 tap-GEMM kernels may run as ONE workgroup
                                                per CU over equal shares of (tile, 32-channel chunk) units instead of one
                                                tile per workgroup -- for grids that do not divide the CUs (152 tiles, 808
                                                tiles ...).  Launches that share a workspace must be ordered (same
                                                stream).  NULL = one tile per workgroup.  Results differ between the
                                                two forms by fp32 summation order only.                              */
  int64_t sk_ws_bytes;
  /* ---- round 6: the producer's gate backward inside the input-gradient tail (split kernels, DV3_EPI_DGRAD) ----------
   * The output y of an input-gradient launch is dL/d(out) of the layer that PRODUCED this layer's input.  When that
   * producer is a Conv1dGLU / HighwayConv1d whose output has no other consumer, its gate backward (autograd of
   * modules.py:157-164, 224-226; stand-alone: dv3_gate_bwd_f32) runs here, on the tile the tail already holds:
   *   pg       the producer's saved pre-gate pair [B][2M][Tout] (its `ab`), fp32; NULL = plain tail
   *   pg_x     HIGHWAY: the producer's input [B][M][Tout] (pg_x_bs / pg_x_rs element strides)
   *   dpg      out: the producer's pre-gate gradient [B][2M][Tout] (what gate_bwd writes as `dab`)
   *   dpg_res  out, HIGHWAY: the gradient of its skip path [B][M][Tout] (`dres`); GLU: the skip gradient is
   *            sqrt(.5) * y, which the producer's own DGRAD launch reads through r / r_scale as before
   *   pg_part  out: bias partial sums, TRANSPOSED [2M][n_part], n_part = ceil(B * Tout / 32): entry (row, k) = the sum
   *            of dpg[row] over the flat (b, t) columns [32 k, 32 k + 32) -- deterministic, summed by
   *            dv3_weight_norm_bwd_f32 (bias_part_t = 1)
   *   pg_mode  DV3_EPI_GLU or DV3_EPI_HIGHWAY; pg_residual as the producer's `residual`
   *   pg_pair  1: dpg is written as PAIR WORDS (below) instead of fp32
   * y itself is written as always (it is the gradient autograd is handed).
   * PAIR WORDS: a 32-bit word (bf16_rn(v) << 16) | bf16_rn(v - bf16_rn(v)) in the place of the fp32 value v -- the two
   * bf16 operands the gradient GEMMs (this entry point in DGRAD form, dv3_wgrad_gemm_f32) would otherwise build from v
   * while staging, computed once by the producer of the tensor.  Same shape, strides and bytes as the fp32 tensor.
   *   x_pair   1: x holds pair words (split_terms == 3 or 0 with a_split: the bf16-pair kernels; no xmask)        */
  const float* pg; const float* pg_x; int64_t pg_x_bs, pg_x_rs;
  float* dpg; float* dpg_res; float* pg_part;
  int32_t pg_mode, pg_residual, pg_pair, x_pair;
} dv3_conv_desc;
#define DV3_IO_IN_BF16 1
#define DV3_IO_OUT_BF16 2
#define DV3_IO_AB_BF16 4   /* only the saved pre-gate pair `ab` is bf16 (y stays fp32): halves the largest tensor a
                              training forward writes; dv3_gate_bwd_desc.ab_bf16 reads it back */
/*
 * Channel-blocked bf16 activation storage ("c8", BASELINE configs 3/4: bf16 activations in HBM, fp32 accumulate):
 * a (B, C, T) activation is held as  bf16 [B][C8][T][8],  C8 = round_up(C,32)/8 -- channel c of frame t is element
 * c%8 of the 16-byte unit (b, c/8, t); channels >= C are zero.  This IS plane 0 of dv3_split_planes_f32's layout: a
 * tensor written by one layer's epilogue is the next layer's `x_planes` (split_terms == 1) with no conversion, the
 * tap-GEMM stages it with plain 16-byte copies, and the epilogue's accumulator tile maps to whole 8-byte halves of
 * units (csrc/conv_common.h).  Any C: the kernels write whole valid groups and keep the padding channels zero;
 * gated layers need Cg % 8 == 0 (the a and gate halves of the saved pre-gate pair start on group boundaries).
 *   DV3_IO_OUT_C8   y (and ab) are written in c8; r / r2 (and the DGRAD addend) are READ in c8
 * (the tensors on the two sides of an epilogue share one layout; x is c8 exactly when x_planes is given).
 */
#define DV3_IO_OUT_C8 16
int dv3_conv_gemm_f32(const dv3_conv_desc* d, void* stream);
/* size in bytes of the stream-K workspace (dv3_conv_desc.sk_ws) on the current device: one accumulator image per CU
 * (128 registers x 512 threads x 4 bytes) behind a 64-byte flag slot per CU                                         */
int dv3_conv_streamk_ws_bytes(void);

/* fp32 (B,C,T) <-> c8 converters (stack entry / exit and their gradients); x strides in elements.            */
int dv3_to_c8_f32(const float* x, int64_t x_bs, int64_t x_rs, uint16_t* out, int32_t B, int32_t C, int32_t T, void* stream);
int dv3_from_c8_f32(const uint16_t* x, float* out, int64_t out_bs, int64_t out_rs, int32_t B, int32_t C, int32_t T,
                    void* stream);
/* the first C channels of a c8 tensor that holds c8p groups per batch item (the `a` half of a pre-gate gradient:
 * the per-frame speaker-bias gradient of a multi-speaker Conv1dGLU, modules.py:157-160)                     */
int dv3_from_c8_head_f32(const uint16_t* x, int32_t c8p, float* out, int64_t out_bs, int64_t out_rs, int32_t B,
                         int32_t C, int32_t T, void* stream);
/* dropout keep-bits [B*C][rs words] (dv3_dropout_bits) -> keep-bytes [B][C8][T] for c8 consumers             */
int dv3_mask_bits_to_c8(const uint32_t* bits, int32_t bits_rs, uint8_t* out, int32_t B, int32_t C, int32_t T, void* stream);

/*
 * Split-bf16 ("bf16x3") operand form.  The fp32 matrix cores run at 1/16 of the bf16 rate on
 * gfx950, so the GEMM kernels can instead write every fp32 operand as hi + lo with
 *   hi = bf16_rn(v), lo = bf16_rn(v - hi)        (|v - hi - lo| <= 2^-18 |v|)
 * and accumulate  A_lo*B_hi + A_hi*B_lo + A_hi*B_hi  in fp32 on v_mfma_f32_32x32x16_bf16:
 * three bf16 MFMAs per product block = 16/3 x the fp32-MFMA rate; the dropped terms bound the
 * relative error of a dot product by ~3*2^-18 of sum|a||b| -- inside the 1e-4 relative parity
 * bar of the reference comparison (tests/ measure it against fp64).
 *
 * dv3_split_pack_bf16 converts a packed fp32 weight image [J][K][lda] (dv3_weight_norm_pack_f32's
 * fwd_pack / bwd_pack) into the layout the bf16x3 tap-GEMM stages with straight 16-byte copies:
 *   out[plane][j][k8][m][8]   plane 0 = hi, 1 = lo; k8 = k/8 over Kp = round_up(K,32) (zero rows
 *   beyond K); m < lda; 8 consecutive k per 16-byte unit (one MFMA A-fragment lane).
 * `out` holds 2*J*Kp*lda uint16 elements.
 *
 * Scaled split-fp16 ("f16x3", dtype = DV3_SPLIT_DTYPE_F16): the same images with
 *   a = v * 2^s, hi = fp16_rn(clamp(a, +-65504)), lo = fp16_rn(a - hi)      (|a - hi - lo| <= 2^-23 |a|
 *   while |a| >= 2^-3; below that the error is absolute, <= 2^-25 in a-units)
 * s = DV3_F16_WEIGHT_SHIFT (8) for weights, DV3_F16_ACT_SHIFT (4) for activations (fixed powers of two:
 * exact, no amax pass; weight-normed |w| <= |g| and O(1..100) activations sit far inside the range);
 * the accumulators carry 2^12 x the result and the epilogue multiplies by 2^-12 (exact).  The forward
 * tap-GEMM uses this form: at the preset model sizes the bf16 split's 2^-17 operands, amplified ~100x by
 * the depth of the network, exceed the 1e-4 parity bar (tests/test_gpu_preset_scale.py measures both);
 * the gradient GEMMs keep the bf16 split, whose exponent range covers gradients without a scale search.
 */
/* Range guard of the f16x3 mode.  The scale shifts are fixed, so |x| > 65504 / 2^4 = 4094 (activations) or
 * |w| > 65504 / 2^8 = 255.9 (weights) leaves the fp16 range.  Nothing is saturated silently: lo is the fp16 of the
 * UNCLAMPED residual, so the pair stays fp16-accurate up to twice the range and turns Inf / NaN beyond it (a NaN or
 * Inf input propagates as in the fp32 reference); and every kernel that builds such pairs counts the 16-byte units
 * that left the range in a sticky device counter.  dv3_f16_range_events copies the counter into dst (device int32,
 * may be NULL) and optionally clears it, both asynchronously on `stream`; the host side (ops.f16_range_events,
 * Trainer.step's scalars) uses it to move a model to the bf16x3 mode, whose operands have fp32's exponent range.
 * Replaces: nothing in the reference (its fp32 kernels have the range of fp32); deepvoice3_pytorch/modules.py:145-164
 * is the computation guarded. */
int dv3_f16_range_events(int32_t* dst, int32_t reset, void* stream);

/* Order two HIP streams: everything enqueued on `to` after this call waits for everything enqueued on `from` before
 * it (hipEventRecord on `from` + hipStreamWaitEvent on `to`, with an event from a ring the library owns; inside a
 * stream capture the pair becomes a graph dependency).  The host-side step uses it to fork the weight-gradient
 * branch of backward onto its own stream and to join it again (ops.SideStream) in one call instead of two torch
 * event calls per layer.  Replaces: nothing in the reference (single stream; train.py:755 loss.backward()). */
int dv3_stream_fork(void* from, void* to);

/* The weight-gradient branch of a CAPTURED step as its own hipGraphs.  A whole-step hipGraph (forward + backward on two
 * streams joined by dv3_stream_fork) replays with the two branches serialised: measured 4-7 % slower than eager launches
 * whenever the GPU is the bound (profiles/r03_side_stream_ab.txt, r04_three_graph_probe.txt).  The caller
 * (train_step.GraphedTrainer) therefore cuts the step into segments: while the step stream is being captured
 * (torch.cuda.CUDAGraph), the side stream is captured SEPARATELY (relaxed mode) through these entry points, and both
 * captures are closed every few layers.  Replay: step-stream segment j, an ordinary event, side segment j on the real
 * second stream -- host-issued hipEventRecord / hipStreamWaitEvent only.  (Event-record / event-wait NODES between two
 * graphs were tried first: bit-identical at small sizes, stale waits -- NaN -- at the benchmark's)
 * Replaces: nothing in the reference.
 *   dv3_graph_side_begin  begin the side stream's own capture
 *   dv3_graph_side_end    end it and instantiate; *n_nodes_out = nodes captured (0: *exec_out is NULL)
 */
int dv3_graph_side_begin(void* side_stream);
int dv3_graph_side_end(void* side_stream, void** exec_out, int32_t* n_nodes_out);
int dv3_graph_launch(void* exec, void* stream);
int dv3_graph_destroy(void* exec);

/* ABI 43: the two branches of a captured backward ordered by a DEVICE flag instead of segment boundaries.  With the
 * segments above every fork-point dependency costs two graph launches and an event pair, the step queue idles 15-40 us at
 * every boundary and side segment j cannot start before step segment j has run to its END.  Here backward is ONE graph
 * per stream: at fork point j the step stream's capture holds a one-thread kernel that publishes (epoch, j) in `flag`, the
 * side stream's capture a one-thread kernel that waits until the flag has reached (epoch, j) -- the weight gradient of a
 * layer starts when its operands exist, whatever else the step stream still has to do.
 *   flag   device uint64, zero before the first use; value = epoch * 4096 + j, monotonic over the steps
 *   epoch  device uint64 OWNED BY THE STREAM that launches the kernel (one for the step stream, one for the side stream),
 *          zero before the first use; `bump` != 0 (the first fork point of a step) increments it first -- both graphs are
 *          replayed once per step, so the two epochs agree
 *   j      fork point of the step, 1 .. 4095
 *   err    device uint32 counting waits that gave up after `timeout_ms` (a lost signal must not hang the GPU); the host
 *          reads it when it reads the step's scalars
 * Release / acquire at agent scope: what the step stream wrote before the signal is visible to the side stream's kernels
 * after the wait (kernel boundaries write back and invalidate as they always do).  The two streams must run on different
 * hardware queues (ops.concurrent_stream probes for that).  Replaces: nothing in the reference. */
int dv3_flag_signal(uint64_t* flag, uint64_t* epoch, int32_t j, int32_t bump, void* stream);
int dv3_flag_wait(const uint64_t* flag, uint64_t* epoch, int32_t j, int32_t bump, uint32_t* err, int32_t timeout_ms,
                  void* stream);

#define DV3_SPLIT_DTYPE_BF16 0
#define DV3_SPLIT_DTYPE_F16 1
#define DV3_SPLIT_F16X3 19
#define DV3_F16_WEIGHT_SHIFT 8
#define DV3_F16_ACT_SHIFT 4
int dv3_split_pack_bf16(const float* packed, uint16_t* out, int32_t J, int32_t K, int32_t lda,
                        int32_t dtype, void* stream);

/*
 * Operand planes of an activation tensor: what the tap-GEMM's B operand looks like after the split, written
 * ONCE by whoever produces the tensor instead of being re-derived by every consuming workgroup.
 *   out[plane][b][c8][t][8]  plane 0 = hi, 1 = lo; c8 < C8p = round_up(C,32)/8 (zero units beyond C);
 *   one 16-byte unit = 8 consecutive channels of one (b, t) column = one MFMA B-fragment lane.
 *   value = x * keep(mask bit) * scale, then  dtype BF16: hi = bf16_rn(v), lo = bf16_rn(v - hi)
 *                                             dtype F16:  a = clamp(v * 2^DV3_F16_ACT_SHIFT), hi/lo = fp16 split
 * `out` holds 2*B*C8p*T*8 uint16.  mask: dropout keep-bits [B*C rows][mask_rs words] or NULL.
 */
typedef struct dv3_planes_desc {
  const float* x; int64_t x_bs, x_rs;        /* [B][C][T]                                   */
  const uint32_t* mask; int32_t mask_rs;
  float scale;                               /* 1/(1-p) of the consuming layer's dropout, or 1 */
  uint16_t* out;
  int32_t B, C, T, dtype;
} dv3_planes_desc;
int dv3_split_planes_f32(const dv3_planes_desc* d, void* stream);

/*
 * dv3_wgrad_gemm_f32 -- weight-gradient GEMM (autograd of F.conv1d w.r.t. weight; also
 * torch.bmm(p, values) of deepvoice3.py:167 when used batched with J=1).
 *
 *   out[s][j][m][c] = sum_{b in slab s} sum_t g[b][m][t] * xd[b][c][t + j*dil - padL]
 *
 * slab s covers batches {s, s+S, s+2S, ...}; S = n_slabs (S == B gives a per-batch
 * result, S == 1 a full reduction).  The slabs are summed by dv3_weight_norm_bwd_f32.
 */
typedef struct dv3_wgrad_desc {
  const float* g;  int64_t g_bs, g_rs;       /* [B][M][T]                                    */
  const float* x;  int64_t x_bs, x_rs;       /* [B][Cin][Tin]                                */
  const uint32_t* xmask; int32_t xmask_rs;   /* dropout keep-bits over x rows, or NULL       */
  float drop_scale;
  float* out;      int64_t out_ss;           /* [S][J][M][ldo]; slab stride                  */
  int32_t ldo;
  int32_t B, M, Cin, T, Tin, J, dil, padL, n_slabs;
  int32_t split_bf16;                        /* 0: exact fp32 MFMA; 1: bf16x3 split-operand MFMA;
                                                2: single-term bf16 MFMA (hi planes only)        */
  int32_t k_split;                           /* 0: slab s = batch items s, s+S, ... (S == B: a per-batch
                                                result); 1 (split-bf16 kernels): slab s = the s-th
                                                contiguous range of the (batch item, 32-step chunk)
                                                sequence -- any S, so the grid can match the chip  */
  int32_t c8;                                /* g and x are channel-blocked bf16 tensors (DV3_IO_OUT_C8 layout, over M and
                                                Cin channels; the stride fields are unused): the bf16-storage form --
                                                split_bf16 == 2, k_split, T == Tin, J in {1, 3}                     */
  const uint8_t* xmask_c8;                   /* c8: dropout keep-bytes over x [B][round_up(Cin,32)/8][Tin], or NULL   */
  int32_t g_pair;                            /* 1 (split_bf16 == 1): g holds PAIR WORDS (dv3_conv_desc: pg_pair), staged
                                                without conversion                                                     */
} dv3_wgrad_desc;
int dv3_wgrad_gemm_f32(const dv3_wgrad_desc* d, void* stream);

/* ------------------------------------------------------------------------------------
 * Weight normalisation (nn.utils.weight_norm, modules.py:85,100,109) + packing.
 *   v: [O][I][J] (Conv1d/Linear, norm over (I,J) per o)   transposed==0
 *   v: [I][O][J] (ConvTranspose1d, norm over (O,J) per i) transposed==1
 * fwd_pack  [J'][K][lda]  operand of the forward tap-GEMM
 * bwd_pack  [J'][K'][ldb] operand of the DGRAD tap-GEMM (transposed, taps reversed)
 * scale     [O or I] = 1/||v||  (saved for backward; w = g*scale*v)
 * For GLU layers (glu_cg > 0) fwd_pack puts the `a` half at [0,Cg) and the gate half at
 * [a_half, a_half+Cg).
 * ------------------------------------------------------------------------------------ */
typedef struct dv3_wn_desc {
  const float* v; const float* g;            /* g NULL => plain weight (no weight norm)     */
  float* scale;
  float* fwd_pack; int32_t lda; int32_t a_half;
  float* bwd_pack; int32_t ldb;
  int32_t O, I, J, transposed, glu_cg;
  int32_t fwd_dtype;                         /* dv3_weight_norm_split_pack_bf16: DV3_SPLIT_DTYPE_* of the
                                                FORWARD image (the input-gradient image is always bf16) */
} dv3_wn_desc;
int dv3_weight_norm_pack_f32(const dv3_wn_desc* d, void* stream);

/* Fused form for the split-bf16 GEMM modes (Conv1d / Linear layers): scale + BOTH split images
 * (dv3_split_pack_bf16 layout; fwd over K = I with lda columns, bwd over K = O with ldb columns,
 * taps reversed) straight from v, g -- no fp32 images.  d->fwd_pack / d->bwd_pack are ignored;
 * bwd_split may be NULL.  K pad rows are written as zeros; pad COLUMNS are left untouched (they
 * only feed output rows the GEMM epilogue drops).                                          */
int dv3_weight_norm_split_pack_bf16(const dv3_wn_desc* d, uint16_t* fwd_split, uint16_t* bwd_split,
                                    void* stream);

/* The same for EVERY weight-normed Conv1d / Linear layer of a model in two launches.  `table_dev` (device memory,
 * n_layers entries, each a dv3_wn_desc as above + its two image pointers), `first_row_dev[l]` = sum of O over the
 * layers before l, `first_block_dev[l]` = sum of ceil(O/32)*ceil(I/32) before l (both int32, device memory);
 * max_taps = max J.  The table holds raw pointers: build it once over fixed buffers (the trainer's flat parameter
 * arena) and reuse it every step -- the call itself reads no host memory, so it can sit inside a captured graph. */
typedef struct dv3_wn_multi_entry {
  dv3_wn_desc d;
  uint16_t* fwd_split; uint16_t* bwd_split;
} dv3_wn_multi_entry;
int dv3_weight_norm_split_pack_multi(const dv3_wn_multi_entry* table_dev, const int32_t* first_row_dev,
                                     const int32_t* first_block_dev, int32_t n_layers, int32_t total_rows,
                                     int32_t total_blocks, int32_t max_taps, void* stream);

/*
 * Per-frame speaker biases of a BLOCK of Conv1dGLU layers in one launch (forward) / three (backward).
 * Replaces, for every Conv1dGLU of a multi-speaker model in training, modules.py:158-162
 *     softsign = F.softsign(self.speaker_proj(speaker_embed)); a = a + softsign
 * (speaker_proj = weight-normed Linear(speaker_embed_dim -> C), modules.py:135) and its autograd, where speaker_embed
 * is the block's expanded and PER-FRAME dropped embedding (deepvoice3.py:78-81, 292-294): one tensor e (B, E, T) for all
 * layers of the block.  forward: layer[l].out (B, C_l, T) = softsign(bias_l + W_l e), W_l = g_l v_l / ||v_l|| by rows --
 * the tensor a layer's tap-GEMM then reads through dv3_conv_desc.spk.  backward: from layer[l].dout (the gradient of
 * that tensor, element strides dout_bs / dout_rs) and the saved out: dv_l, dg_l, dbias_l += (weight-norm backward
 * included) and de (B, E, T) = the gradient of e, overwritten.  Sums in a fixed order (bit-identical run to run).
 * `layers` is HOST memory (n_layers <= DV3_SPK_MAX_LAYERS entries, copied into the kernel arguments), E <= 16.
 * Workspace of the backward: dv3_speaker_bias_bwd_scratch_floats() floats.
 */
#define DV3_SPK_MAX_LAYERS 16
typedef struct dv3_spk_layer {
  const float* v; const float* g; const float* bias;   /* (C, E) direction, (C) magnitude or NULL, (C) bias or NULL */
  float* out;                                           /* (B, C, T) fp32, contiguous: written by fwd, read by bwd    */
  const float* dout; int64_t dout_bs; int64_t dout_rs;  /* bwd */
  float* dv; float* dg; float* dbias;                   /* bwd, += */
  int32_t C;
  int32_t dout_c8p;                                     /* bwd: != 0: `dout` is a c8 bf16 tensor [B][dout_c8p][T][8] (the
                                                           pre-gate gradient a c8 layer produced: its first C channels
                                                           are the gradient of this bias); dout_bs / dout_rs unused   */
} dv3_spk_layer;
typedef struct dv3_spk_desc {
  const float* e; int64_t e_bs; int64_t e_rs;           /* (B, E, T) fp32, element strides (frames contiguous)       */
  float* de;                                            /* bwd: (B, E, T) fp32 contiguous                             */
  float* scratch; int64_t scratch_floats;               /* bwd                                                        */
  int32_t B, E, T, n_layers;
} dv3_spk_desc;
int dv3_speaker_bias_fwd_f32(const dv3_spk_desc* d, const dv3_spk_layer* layers, void* stream);
int dv3_speaker_bias_bwd_scratch_floats(const dv3_spk_desc* d, const dv3_spk_layer* layers);
int dv3_speaker_bias_bwd_f32(const dv3_spk_desc* d, const dv3_spk_layer* layers, void* stream);

/*
 * Backward of weight norm from wgrad slabs: dW = sum_s slab[s]; dg, dv.
 *   slabs: [S][J][M][ldo] as written by dv3_wgrad_gemm_f32 (M = O rows; c = I cols;
 *   for transposed (ConvTranspose) layers M = J*O rows and the tap is folded in m).
 *   bias_part: [P][O] partial bias sums (or NULL), reduced into dbias.
 */
typedef struct dv3_wn_bwd_desc {
  const float* slabs; int64_t slab_ss; int32_t ldo; int32_t n_slabs;
  const float* v; const float* g; const float* scale;
  float* dv; float* dg;                      /* dg NULL when g NULL (plain weight: dv = dW) */
  const float* bias_part; int32_t n_part; float* dbias;
  int32_t O, I, J, transposed;
  int32_t accumulate;                        /* 1: dv, dg, dbias += (gradient buffers that are zeroed once
                                                per step and shared by every use of the parameter);
                                                0: overwrite                                          */
  int32_t bias_part_t;                       /* 0: bias_part is [n_part][O]; 1: [O][n_part] (dv3_conv_desc.pg_part) */
} dv3_wn_bwd_desc;
int dv3_weight_norm_bwd_f32(const dv3_wn_bwd_desc* d, void* stream);
/* The same for n <= DV3_WN_BWD_MULTI_MAX layers in ONE launch: one workgroup per normalised row of every layer.  A
 * layer's launch is a chain of ~24 memory round trips on 2 workgroups per CU (36 us whatever the layer size); several
 * layers together fill the chip and overlap those chains.  The descriptors (HOST memory) travel by value as kernel
 * arguments: no table upload.  Two entries must not write the same gradient buffers.  Bit-identical to n single calls.
 * Replaces: autograd of nn.utils.weight_norm for the layers of loss.backward() (train.py:755). */
#define DV3_WN_BWD_MULTI_MAX 8
int dv3_weight_norm_bwd_multi(const dv3_wn_bwd_desc* descs, int32_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * Gate / activation backward (autograd of modules.py:157-164, 224-226 and of ReLU/sigmoid).
 *   mode GLU/HIGHWAY: in dy [B][Cg][T], ab [B][2Cg][T], x (highway input) ->
 *        dab [B][2Cg][T], dres [B][Cg][T] (gradient flowing to the residual / highway
 *        carry; NULL when residual==0), bias_part [B][2Cg] row sums of dab,
 *        dspk [B][Cg] (row sums of d a, = bias_part's first half; written when non-NULL)
 *   mode RELU/SIGMOID/LINEAR: in dy, y [B][M][T] -> dpre [B][M][T], bias_part [B][M]
 * ------------------------------------------------------------------------------------ */
typedef struct dv3_gate_bwd_desc {
  const float* dy; const float* ab_or_y; const float* x;
  float* dab; float* dres; float* bias_part;
  float alpha;                               /* non-gated modes: dy is scaled by alpha first */
  int32_t B, C, T, mode, residual;
  int32_t ab_bf16;                           /* gated modes: ab_or_y is a bf16 tensor (DV3_IO_AB_BF16 / _OUT_BF16)  */
  int32_t c8;                                /* every activation tensor (dy, ab_or_y, x, dab, dres) is channel-blocked
                                                bf16 (DV3_IO_OUT_C8 layout; C % 8 == 0); bias_part stays fp32        */
  int32_t dab_pair;                          /* gated modes, fp32 tensors: dab is written as PAIR WORDS (dv3_conv_desc:
                                                pg_pair) -- the layer's two gradient GEMMs stage it without conversion
                                                (dv3_conv_desc.x_pair, dv3_wgrad_desc.g_pair)                        */
} dv3_gate_bwd_desc;
int dv3_gate_bwd_f32(const dv3_gate_bwd_desc* d, void* stream);

/* ------------------------------------------------------------------------------------
 * Dropout keep-bit generator (F.dropout's bernoulli_, modules.py:147,210; deepvoice3.py:
 * 75,80,165,290,321).  Philox4x32-10, counter = (word index, site), key = seed.
 * bits[w] bit i == 1  <=>  element 32*w+i of the row is kept; p quantised to 1/65536.
 * dev_seed_offset: a device-resident step counter, so a replayed hipGraph draws new masks.
 * ------------------------------------------------------------------------------------ */
int dv3_dropout_bits(uint32_t* bits, int64_t n_words, float p, uint64_t seed,
                     uint64_t site, const uint64_t* dev_seed_offset /* added to seed; NULL = 0 */,
                     void* stream);

/* The same keep decisions as dv3_dropout_bits over rows [B*C] (same seed / site / offset), written as the keep-BYTES
 * [B][round_up(C,32)/8][T] the c8 consumers read (= dv3_dropout_bits + dv3_mask_bits_to_c8 in one launch).      */
/* Both forms of one dropout site in ONE launch: keep-bits [B*C][ceil(T/32)] (what the weight-gradient and
 * input-gradient kernels read) and keep-bytes [B][round_up(C,32)/8][T] (what the 256 x 256 split tap-GEMM and the c8
 * kernels stage: one byte load per 8-channel item instead of eight keep-bit words).  Same decisions as
 * dv3_dropout_bits for the same seed / site / offset.  Replaces F.dropout's bernoulli_ (modules.py:147,210). */
int dv3_dropout_bits_keep(uint32_t* bits, uint8_t* keep, int32_t B, int32_t C, int32_t T, float p, uint64_t seed,
                          uint64_t site, const uint64_t* dev_seed_offset, void* stream);
int dv3_dropout_keep_c8(uint8_t* out, int32_t B, int32_t C, int32_t T, float p, uint64_t seed, uint64_t site,
                        const uint64_t* dev_seed_offset, void* stream);

/* Several dropout sites in ONE launch (ABI 41): site l gets exactly what dv3_dropout_bits_keep(bits, keep, B, C, T, p, seed,
 * site, ...) (bits != NULL) or dv3_dropout_keep_c8(keep, ...) (bits == NULL) writes; keep == NULL with B = 1, C = rows:
 * what dv3_dropout_bits(bits, rows * ceil(T/32), ...) writes.  A training step draws 25-35 masks
 * (one per Conv1dGLU / HighwayConv1d: modules.py:147,210), each a ~6 us launch on the forward's only queue; the host side
 * (ops.MaskPlan) issues them together at the start of the step once the step's list of sites has repeated.        */
#define DV3_DROPOUT_MULTI_MAX 48
typedef struct dv3_dropout_site {
  uint8_t* keep;  uint32_t* bits;      /* keep-bytes [B][round_up(C,32)/8][T]; keep-bits [B*C][ceil(T/32)] or NULL */
  int32_t B, C, T;  float p;
  uint64_t site;
} dv3_dropout_site;
int dv3_dropout_keep_c8_multi(const dv3_dropout_site* sites, int32_t n, uint64_t seed,
                              const uint64_t* dev_seed_offset, void* stream);

/* out[row][t] = x[row][t] * keep(bits,row,t) * scale -- a standalone F.dropout for the few
 * sites whose dropped tensor is shared by several consumers (deepvoice3.py:78-80,290,321:
 * the time-expanded speaker embedding and the decoder input).  Its own backward.          */
int dv3_dropout_apply_f32(const float* x, const uint32_t* bits, int32_t bits_rs, float scale,
                          float* out, int64_t rows, int32_t T, void* stream);

/* ------------------------------------------------------------------------------------
 * Attention (deepvoice3.py:132-176): masked softmax over the key axis + dropout.
 *   s [B][Tq][Tk] scores (in place -> probabilities p, returned PRE-dropout as the
 *   reference does, deepvoice3.py:163); pd (optional) = dropout(p) for the context GEMM.
 *   key_len [B] (int32, use_memory_mask) or NULL; last_attended (device scalar) with
 *   win_back/win_ahead: monotonic window [last-back, last+ahead) (deepvoice3.py:150-156).
 * ------------------------------------------------------------------------------------ */
typedef struct dv3_softmax_desc {
  float* s; float* pd;
  const int32_t* key_len;
  const int32_t* last_attended;              /* device int32[1] or NULL (no window)        */
  const uint32_t* mask; int32_t mask_rs; float drop_scale;
  float pd_scale;                            /* pd = dropout(p) * pd_scale (the sqrt(Tk) of deepvoice3.py:170-171) */
  int32_t B, Tq, Tk, win_back, win_ahead;
  const float* pd_scale_dev;                 /* ABI 42: device float[1] multiplied into pd_scale, or NULL (valid-length
                                              * steps: the key count of the batch is only known on the device)     */
} dv3_softmax_desc;
int dv3_attn_softmax_f32(const dv3_softmax_desc* d, void* stream);
/* backward: ds = p * (dp_total - sum_k(dp_total*p)); dp_total = dpd*bit*scale + dp_direct */
typedef struct dv3_softmax_bwd_desc {
  const float* p; const float* dpd; const float* dp_direct; float* ds;
  const uint32_t* mask; int32_t mask_rs; float drop_scale;
  int32_t B, Tq, Tk;
  const float* scale_dev;                    /* ABI 42: device float[1] multiplied into drop_scale, or NULL          */
} dv3_softmax_bwd_desc;
int dv3_attn_softmax_bwd_f32(const dv3_softmax_bwd_desc* d, void* stream);
/* Fused attention forward (deepvoice3.py:143-171, teacher-forced / training: no monotonic window): scores = q^T k
 * (exact fp32 MFMA) -> -inf beyond key_len[b] -> softmax -> P; pd = P * keep(mask) * drop_scale * pd_scale;
 * ctx[b][e][t] = sum_n v[b][e][n] * pd[b][t][n] (exact fp32 MFMA), one launch.  q (B,E,Tq), k (B,E,Tk) BCT;
 * vT (B,Tk,E) = the values in the reference's own (B,T,C) layout; P, pd (B,Tq,Tk) are both written once (P is the
 * alignment the layer returns, both are what dv3_attn_softmax_bwd_f32 and the gradient products read).  Tk <= 511. */
typedef struct dv3_attn_fwd_desc {
  const float* q; const float* k; const float* vT;
  const int32_t* key_len;
  const uint32_t* mask; int32_t mask_rs; float drop_scale;
  float pd_scale;
  float* ctx; float* P; float* pd;
  int32_t B, E, Tq, Tk;
  const float* pd_scale_dev;                 /* ABI 42: as dv3_softmax_desc.pd_scale_dev                             */
} dv3_attn_fwd_desc;
int dv3_attn_fwd_f32(const dv3_attn_fwd_desc* d, void* stream);
/* argmax over keys of one probability row -> last_attended (deepvoice3.py:445)          */
int dv3_attn_argmax_i32(const float* p_row, int32_t Tk, int32_t* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Small layout / elementwise helpers.
 * ------------------------------------------------------------------------------------ */
/* y[b][c][r] = alpha * x[b][r][c] (+ beta_add[b][c][r] if non-NULL) : BTC <-> BCT      */
int dv3_transpose_f32(const float* x, float* y, const float* add, int32_t B, int32_t R,
                      int32_t C, float alpha, void* stream);
/* out[i] = alpha * (a[i] + b[i]) ; b may be NULL                                       */
int dv3_axpby_f32(const float* a, const float* b, float* out, int64_t n, float alpha,
                  void* stream);
/* out[0] = a[0] + b[0] (+ c[0]) (+ d[0]): the total of the loss terms (train.py:728-740); c, d may be NULL  */
int dv3_sum_scalars_f32(const float* a, const float* b, const float* c, const float* d, float* out, void* stream);
/* bytes of device memory to `value` -- a fill KERNEL on the caller's stream, not hipMemsetAsync (a captured memset node
 * replayed with a corrupt pattern: DESIGN.md 3.7): the gradient arena before backward (optimizer.zero_grad(),
 * train.py:683)                                                                                                 */
int dv3_memset_b8(void* p, int32_t value, int64_t bytes, void* stream);
/* `rows` runs of row_bytes each, row_stride_bytes apart (all multiples of 16, p 16-byte aligned): the padding groups of a
 * channel-blocked tensor whose channel count is not a multiple of 32 (ops._c8_empty), one launch.                    */
int dv3_memset_rows_b8(void* p, int32_t value, int64_t rows, int64_t row_bytes, int64_t row_stride_bytes, void* stream);
/* ABI 42 (valid-length steps).  x is [rows][T] elements of `words` 32-bit words each (fp32 (B, C, T): words = 1; the
 * channel-blocked bf16 [B][C8][T][8]: words = 4).  Columns t >= t_valid[0] * mult of every row are set to zero, where
 * t_valid is a device int32[1]; the host only promises T - t_valid[0] * mult <= max_tail.  What a non-causal
 * convolution of the reference sees beyond the batch's longest item is its own zero padding (modules.py:139-143:
 * nn.Conv1d(padding=...)): a batch padded further (to a lattice shape, so that captured steps can be replayed) keeps
 * that by zeroing the activations -- and, in backward, the activation gradients -- beyond the batch's own maximum.  */
int dv3_zero_tail_b32(void* x, int64_t rows, int32_t T, int32_t words, const int32_t* t_valid, int32_t mult,
                      int32_t max_tail, void* stream);
/* Embedding gather into BCT with optional dropout: out[b][c][t] = W[idx[b][t]][c]
 * (deepvoice3.py:74-75, nyanko.py:63).  Backward: dense scatter-add into dW.           */
int dv3_embedding_bct_f32(const int64_t* idx, const float* w, float* out,
                          const uint32_t* mask, int32_t mask_rs, float drop_scale,
                          int32_t B, int32_t T, int32_t C, int32_t n_vocab, void* stream);
int dv3_embedding_bct_bwd_f32(const int64_t* idx, const float* dout, float* dw,
                              const uint32_t* mask, int32_t mask_rs, float drop_scale,
                              int32_t B, int32_t T, int32_t C, int32_t n_vocab,
                              int32_t padding_idx, void* stream);
/* SinusoidalEncoding.forward (modules.py:30-64) at gathered positions only, BCT output:
 * out[b][c][t] = base[b][c][t] + enc, enc = (pos==0 || !apply_sincos) ? w*table[pos][c]
 *              : (c odd ? cos : sin)(w*table[pos][c]);  w = w[b] (w_per_batch) or w[0] or 1 */
int dv3_sincos_pos_bct_f32(const int64_t* pos, const float* table, const float* w,
                           int32_t w_per_batch, const float* base, float* out, int32_t B,
                           int32_t T, int32_t C, int32_t n_pos, int32_t apply_sincos,
                           void* stream);
/* gradient w.r.t. the TABLE (trainable_positional_encodings=True, deepvoice3_pytorch/__init__.py:53-57): dtable
 * [n_pos][C], row 0 (padding) zero; w NULL = rate 1; apply_sincos 0 = plain embedding rows (nyanko.py:162-169) */
int dv3_sincos_pos_table_bwd_f32(const int64_t* pos, const float* table, const float* w, int32_t w_per_batch,
                                 const float* dout, float* dtable, int32_t B, int32_t T, int32_t C,
                                 int32_t n_pos, int32_t apply_sincos, void* stream);
/* gradient of the encoding w.r.t. the rate: dw[b] = sum_{c,t} dout[b][c][t] * d enc/d w
 * (multi-speaker models learn the rate through speaker_proj1/2, deepvoice3.py:304-315).   */
int dv3_sincos_pos_bwd_f32(const int64_t* pos, const float* table, const float* w,
                           int32_t w_per_batch, const float* dout, float* dw, int32_t B, int32_t T,
                           int32_t C, int32_t n_pos, void* stream);
/* The same in two deterministic stages (round 6; ABI 40): `n_chunks` workgroups per batch item leave partial sums in
 * `partial` [B][n_chunks], a second launch adds each item's in index order -- the one-workgroup-per-item form above is
 * 64 workgroups of 200 dependent sinf / cosf iterations on a 256-CU part (160 us per call in the deepvoice3_vctk step). */
int dv3_sincos_pos_bwd2_f32(const int64_t* pos, const float* table, const float* w,
                            int32_t w_per_batch, const float* dout, float* partial, int32_t n_chunks, float* dw,
                            int32_t B, int32_t T, int32_t C, int32_t n_pos, void* stream);
/* Incremental-conv window (conv.py:34-46): buf [rows][L] shifts left one frame and takes
 * x[row * x_stride] as its newest; in place, static addresses (hipGraph-capturable decode step) */
int dv3_shift_append_f32(float* buf, const float* x, int64_t rows, int32_t L, int64_t x_stride,
                         void* stream);
/* Device-side batch padding (the `_pad` / `_pad_2d` loops of train.collate_fn, train.py:293-360, and the
 * mel time down-sampling of train.py:639-640 in the same pass): items are packed back to back as rows of
 * D 32-bit words (f32 features; int64 text ids as D = 2), item b owning rows [row_off[b], row_off[b+1]).
 *   out[b][t][0..D) = src[row_off[b] + s][0..D),  s = t * t_stride - lead,   if 0 <= s < rows of item b
 *                   = 0                                                      otherwise
 * out is [B][T_out][D].  lead = the b_pad leading zero frames, t_stride = downsample_step for the mel. */
int dv3_ragged_pad_rows_b32(const uint32_t* src, const int32_t* row_off, uint32_t* out, int32_t B,
                            int32_t T_out, int32_t D, int32_t lead, int32_t t_stride, void* stream);
/* dy [B][O][2T] -> out[b][j*O+o][t] = dy[b][o][2t+j]: operand of the ConvTranspose1d(k2,s2)
 * backward GEMMs (deepvoice3.py:519-520,527-528)                                          */
int dv3_deinterleave2_f32(const float* dy, float* out, int32_t B, int32_t O, int32_t T,
                          void* stream);

/* ------------------------------------------------------------------------------------
 * Losses (train.py:261-291,537-601,704-740), fused value + gradient.
 * ------------------------------------------------------------------------------------ */
/* spec_loss on y_hat[:, :-r] vs y[:, r:]  (BTC tensors [B][T][D]).
 *   loss = (1-wbd)*(wm*maskedL1 + (1-wm)*L1) + wbd*(wm*masked_mean(z) + (1-wm)*mean(z))
 *   z = -y*logit(yh) + log1p(exp(logit(yh))), logit eps 1e-8 (train.py:537-582).
 * out4 = {l1_loss, binary_div, total, mask_sum}; dyh gets d total / d y_hat * gscale.   */
typedef struct dv3_spec_loss_desc {
  const float* y_hat; const float* y; const int32_t* lengths; /* mask: t < lengths[b] (shifted by r) */
  float* dyh; float* out4; float* scratch;  /* scratch: >= 4*n_blocks floats                */
  int64_t yh_bs, yh_ts, yh_ds;              /* element strides of y_hat / dyh over (b,t,d):  */
  int64_t y_bs, y_ts, y_ds;                 /* BTC (T*D, D, 1) or BCT (D*T, 1, T) both fine  */
  int32_t B, T, D, r; float w_masked, w_bd, gscale;
  const int32_t* t_valid;                   /* ABI 42: device int32[1] or NULL.  The loss is the one of the tensors cut
                                             * to their first t_valid[0] frames (means over B * (t_valid - r) * D, zero
                                             * gradient beyond): a batch padded beyond its own maximum (valid-length
                                             * steps, replayed from a lattice of padded shapes)                         */
} dv3_spec_loss_desc;
int dv3_spec_loss_f32(const dv3_spec_loss_desc* d, void* stream);
int dv3_spec_loss_scratch_floats(int32_t B, int32_t T, int32_t D);

/* guided attention (train.py:585-601,733-740): loss = mean(attn * W),
 * W[b][t][n] = 1-exp(-(n/N_b - t/T_b)^2/(2 g^2)) inside (T_b,N_b), 0 outside.
 * attn [L][B][Tq][Tk]; dattn (optional) = W/(L*B*Tq*Tk) * gscale                        */
int dv3_guided_attn_loss_f32(const float* attn, const int32_t* in_len, const int32_t* out_len,
                             float* dattn, float* out1, float* scratch, int32_t L, int32_t B,
                             int32_t Tq, int32_t Tk, float g, float gscale, void* stream);
/* ABI 42: the same with the mean taken over L * B * tq_valid[0] * tk_valid[0] elements (device int32[1] each): the
 * alignment of a batch padded beyond its own maxima.  W is zero outside every item's (T_b, N_b), so only the divisor
 * differs from the call above on the padded tensor.                                                               */
int dv3_guided_attn_loss_valid_f32(const float* attn, const int32_t* in_len, const int32_t* out_len,
                                   float* dattn, float* out1, float* scratch, int32_t L, int32_t B,
                                   int32_t Tq, int32_t Tk, float g, float gscale, const int32_t* tq_valid,
                                   const int32_t* tk_valid, void* stream);
/* BCELoss(done_hat, done) mean (train.py:614,714) + gradient                            */
int dv3_bce_loss_f32(const float* p, const float* t, float* dp, float* out1, float* scratch,
                     int64_t n, float gscale, void* stream);
/* ABI 42: p, t are [rows][T]; only the first t_valid[0] (device int32[1]) columns of every row take part: mean over
 * rows * t_valid, zero gradient beyond.                                                                             */
int dv3_bce_loss_valid_f32(const float* p, const float* t, float* dp, float* out1, float* scratch,
                           int64_t rows, int32_t T, const int32_t* t_valid, float gscale, void* stream);

/* ------------------------------------------------------------------------------------
 * Optimiser tail (train.py:755-759): clip_grad_norm_ + Adam over ONE flat fp32 arena
 * (all trainable parameters / gradients / moments are views into four flat buffers).
 *   out2 = {||g||_2, ||g||_2^2};  hyper (device) = {lr, 1-beta1^t, sqrt(1-beta2^t)};
 *   grad_prescale multiplies g first (1/world_size after a sum all-reduce).
 * ------------------------------------------------------------------------------------ */
int dv3_grad_sqnorm_f32(const float* g, int64_t n, float* partial, int32_t n_partial,
                        float* out2, void* stream);
int dv3_clip_adam_f32(float* p, const float* g, float* m, float* v, int64_t n,
                      const float* grad_norm, float clip, const float* hyper, float beta1,
                      float beta2, float eps, float weight_decay, float grad_prescale,
                      void* stream);

/* ------------------------------------------------------------------------------------
 * Autoregressive decode step (Decoder.incremental_forward, deepvoice3.py:397-473): two fused kernels per
 * layer kind instead of the ~100 launches of the module-by-module path.  Both read the 0-based step counter
 * `t` from DEVICE memory, so one captured hipGraph of a step replays for every step.
 *
 * dv3_conv_step_f32 -- one incremental conv layer (conv.py:17-46) with its whole tail:
 *   window: tap J-1 = the new frame x (B, Cin); tap j = the frame (J-1-j)*dil steps back, kept in `ring`
 *   [L][B][Cin] (slot t mod L holds step t; the call stores x there; L >= (J-1)*dil + 1; zero the ring to
 *   start a sequence = conv.py clear_buffer);  acc[b][m] = sum_{j,c} a[j][c][m] * window[b][j][c]
 *   (a = the layer's weights in STEP-TILE order, written by dv3_conv_step_pack_f32 from dv3_weight_norm_pack_f32's
 *   fwd_pack: [row block of 16][window element j*Cin + c][16 rows], gated layers [..][16 `a` rows | 16 gate rows] --
 *   the 100 KB a workgroup streams per layer are one dense block; lda / a_half are ignored);  then, by mode:
 *     GLU / HIGHWAY  (+bias, +spk[b][m] on the `a` half) gate with the new frame as residual / highway carry
 *                    (modules.py:157-164, 224-226), then r2: y = (y + r2) * sqrt(.5)
 *     LINEAR / RELU / SIGMOID / SOFTSIGN  activation, then r and r2 residuals, each (y + r) * sqrt(.5)
 *   then y += post_add[t*post_add_ts + b*post_add_bs + m] (the step's position encoding, deepvoice3.py:430);
 *   y -> (B, Cout); y_act (optional) = sigmoid(y); out_seq (optional) [t][b][m] = y_act if given else y.
 * dv3_attn_step_f32 -- one attention read (deepvoice3.py:143-171 at Tq = 1): scores q.k over the window
 *   [last-win_back, last+win_ahead) (all keys when last_attended is NULL), softmax, ctx = P v * Tk*sqrt(1/Tk);
 *   attn (B, Tk) and / or attn_seq [t][b][n]; last_attended is a PAIR of device ints: slot t&1 is read, the
 *   argmax of batch item 0 (deepvoice3.py:445) is stored in slot (t+1)&1.
 * ------------------------------------------------------------------------------------ */
typedef struct dv3_conv_step_desc {
  const float* x; int64_t x_bs;
  float* ring; int32_t L;
  const int32_t* t;
  const float* a; int32_t lda, a_half;
  const float* bias;
  const float* spk; int64_t spk_bs;
  const float* r;  int64_t r_bs;
  const float* r2; int64_t r2_bs;
  const float* post_add; int64_t post_add_ts, post_add_bs;
  float* y; int64_t y_bs;
  float* y_act; int64_t y_act_bs;
  float* out_seq; int64_t out_seq_ts, out_seq_bs;
  int32_t B, Cin, M, Cg, J, dil, mode, residual;
  float* y_pre; int64_t y_pre_bs;            /* optional: the layer output BEFORE post_add (nyanko.py:297-300: Q feeds the
                                                concat while Q + position code feeds the attention query)            */
  int64_t x_ts;                              /* the new frame of step t is x + t*x_ts (teacher forcing: x = test_inputs,
                                                deepvoice3.py:411-415); 0 = the same buffer every step               */
  int32_t t_value, reserved;                 /* the step index when `t` is NULL (host-driven loops: dv3_decode_program_launch) */
} dv3_conv_step_desc;
int dv3_conv_step_f32(const dv3_conv_step_desc* d, void* stream);
/* fwd_pack ([J*Cin][lda] fp32; gated: `a` rows at column 0, gate rows at column a_half) -> the step-tile image
 * dv3_conv_step_f32 reads.  Cg = gated ? rows per half : 0.  out holds dv3_conv_step_pack_floats(...) floats. */
int dv3_conv_step_pack_floats(int32_t Ktot, int32_t M, int32_t Cg);
/* LDS bytes one dv3_conv_step_f32 launch of a J-tap layer with Cin input channels needs (its k-tap window of the
 * batch group + the two reduction stages); the launch is refused above DV3_CONV_STEP_LDS_MAX.  The Python decoders ask
 * this before they choose the fused step program (deepvoice3.py / nyanko.py: _fast_decode_eligible). */
#define DV3_CONV_STEP_LDS_MAX 65536
int dv3_conv_step_lds_bytes(int32_t J, int32_t Cin);
int dv3_conv_step_pack_f32(const float* fwd_pack, int32_t lda, int32_t a_half, int32_t Ktot, int32_t M, int32_t Cg,
                           float* out, void* stream);

typedef struct dv3_attn_step_desc {
  const float* q; int64_t q_bs;              /* (B, E)                                       */
  const float* k; const float* v;            /* (B, E, Tk) each, BCT (or (B, Tk, E): kv_tke)  */
  int32_t* last_attended;                    /* [2] device ints or NULL                      */
  int32_t win_back, win_ahead;
  const int32_t* t;
  float* ctx; int64_t ctx_bs;                /* (B, E)                                       */
  float* attn;                               /* (B, Tk) or NULL                              */
  float* attn_seq; int64_t attn_seq_ts;      /* [t][B][Tk] or NULL                           */
  int32_t B, E, Tk;
  int32_t t_value;                           /* the step index when `t` is NULL             */
  int32_t kv_tke, reserved;                  /* 1: k and v are (B, Tk, E) -- the reference's own layout (deepvoice3.py:
                                                132-141), a key / value row is contiguous: coalesced for a one-frame read */
} dv3_attn_step_desc;
int dv3_attn_step_f32(const dv3_attn_step_desc* d, void* stream);

/* The whole decoder loop in ONE launch (Decoder.incremental_forward's `while True`, deepvoice3.py:397-473;
 * nyanko.py:277-331).  `entries` is the per-step launch program -- the same descriptors dv3_conv_step_f32 /
 * dv3_attn_step_f32 take, in execution order, in DEVICE memory; their `t` pointers are ignored (the step index is the
 * kernel's loop variable).  A persistent grid of ceil(B/4) batch groups x wg_per_group workgroups walks the program:
 * the workgroups of one batch group share an entry's output-channel blocks and meet at a group barrier after every
 * entry (the activations of 4 batch items never leave their group); all groups meet once per step, where the stop
 * rule of the reference is evaluated on the device:
 *     steps = t + 1;  stop if (done_seq != NULL and steps > min_steps and done_seq[t][b] > 0.5 for all b)
 *                     or steps > max_steps                       (deepvoice3.py:463-470; without done_seq: n_steps)
 * The number of steps executed is stored in steps_out.  `sync` is device scratch of dv3_decode_program_sync_ints(B)
 * int32 that the call zeroes first.  The grid must be co-resident (it is at most 256 workgroups of 256 threads).
 * A barrier that does not complete in ~2 s of spinning sets steps_out to -1 and the kernel exits. */
typedef struct dv3_decode_entry {
  int32_t kind;                              /* 0: conv step, 1: attention step */
  int32_t reserved;
  dv3_conv_step_desc conv;
  dv3_attn_step_desc attn;
} dv3_decode_entry;
typedef struct dv3_decode_program {
  const dv3_decode_entry* entries;           /* device */
  const dv3_decode_entry* entries_host;      /* the host copy `entries` was uploaded from: validated by the call */
  int32_t n_entries, B;
  int32_t t0, n_steps;                       /* steps t0 .. t0 + n_steps - 1 at most */
  const float* done_seq; int64_t done_ts;    /* [t][b] done flags the program writes, or NULL */
  int32_t min_steps, max_steps;
  int32_t* sync;
  int32_t* steps_out;                        /* device int: steps executed in this call */
  int32_t wg_per_group;                      /* 0 = library default */
  int32_t reserved;
} dv3_decode_program;
int dv3_decode_program_sync_ints(int32_t B);
int dv3_decode_program_run(const dv3_decode_program* prog, void* stream);
/* The same program, host driven: steps t0 .. t0 + n_steps - 1 as one kernel launch per entry per step, issued by ONE
 * call (a step of the ljspeech decoder is 17 launches: issued from Python + ctypes or as a replayed per-step hipGraph
 * they cost the host ~110 us per step, more than the GPU needs; from this loop ~3 us each).  Uses entries_host; the
 * step index travels in the descriptors (t_value), no device counter.  No stop rule here: the caller runs a chunk of
 * steps, reads the done flags of the chunk, and discards the steps after the stopping one (later steps never change
 * earlier outputs, so the kept prefix is what a step-by-step loop produces). */
int dv3_decode_program_launch(const dv3_decode_program* prog, void* stream);

/* ------------------------------------------------------------------------------------
 * Audio inverse (audio.py:37-43, synthesis.py:64-71): linear spectrogram -> waveform on the
 * device.  The reference calls the third-party `lws` package for phase reconstruction (not
 * vendored: parity unpinned); these entry points implement Griffin-Lim with
 * torch.stft / torch.istft conventions: n_fft 1024 (513 bins), periodic Hann window, hop as
 * given (preset: 256), center=True with reflect padding, istft normalised by the overlap-added
 * squared window, signal length L = hop*(T-1).
 *   mag     [B][T][513]      magnitudes ** power
 *   phasor  [B][T][513][2]   unit complex phase estimate (re, im)
 *   frames  [B][T][1024]     windowed time-domain frames
 *   y       [B][L]
 * One Griffin-Lim iteration = istft_frames -> overlap_add -> stft_phase, or fused: gl_project -> overlap_add.
 * ------------------------------------------------------------------------------------ */
/* mag = (10^((clip(x,0,1)*(-min_db) + min_db + ref_db)/20))^power   audio.py:39-41,84-93 */
int dv3_gl_prepare_f32(const float* lin, float* mag, int64_t n, float min_level_db,
                       float ref_level_db, float power, void* stream);
int dv3_istft_frames_f32(const float* mag, const float* phasor /* NULL: zero phase */,
                         float* frames, int32_t B, int32_t T, void* stream);
int dv3_overlap_add_f32(const float* frames, float* y, int32_t B, int32_t T, int32_t hop,
                        void* stream);
/* one Griffin-Lim projection, fused: frames = istft_frames(mag, phase(stft(y))) without storing the phasors */
int dv3_gl_project_f32(const float* y, const float* mag, float* frames, int32_t B, int32_t T, int32_t hop,
                       void* stream);
/* outputs (each may be NULL): phasor, spec = the complex STFT [B][T][513][2], mag_bct = |STFT|
 * as [B][513][T] (the channel-major operand of the mel filterbank GEMM)                     */
int dv3_stft_phase_f32(const float* y, float* phasor, float* spec, float* mag_bct, int32_t B,
                       int32_t T, int32_t hop, void* stream);
/* Forward analysis (audio.py:21-23,31-35,46-51,79-89): preemphasis, then dv3_stft_phase_f32
 * (mag_bct), the mel filterbank as a 1x1 tap-GEMM (dv3_conv_gemm_f32, M = num_mels, Cin = 513),
 * then amplitude -> dB -> [0,1] normalisation.                                              */
int dv3_preemphasis_f32(const float* x, float* y, int32_t B, int32_t L, float coef, void* stream);
int dv3_amp_to_db_norm_f32(const float* x, float* out, int64_t n, float min_level_db,
                           float ref_level_db, void* stream);
/* y[n] = x[n] + coef*y[n-1] per row: inv_preemphasis, audio.py:26-28 (nnmnkwii's lfilter([1], [1, -coef])).  x != y
 * runs chunked in parallel over the row (exact to fp32 rounding for |coef| well below 1: each 3072-sample chunk restarts
 * 1024 samples early); x == y, or a coefficient whose memory outlasts the warm-up, runs one workgroup per row. */
int dv3_deemphasis_f32(const float* x, float* y, int32_t B, int32_t L, float coef, void* stream);

/* The same analysis / inverse on the framing of `lws.lws(fft_size, hop_size, mode="speech")` (audio.py:54-55), which the
 * reference uses for its features (`.stft`, audio.py:31-35,46-51) and for the framing of its inverse (`.istft`,
 * audio.py:42) -- the package's published conventions (python/lws.pyx; restated and cited in oracle/audio_oracle.py):
 *   awin  [1024]  analysis window: sqrt of the SYMMETRIC Hann window
 *   swin  [1024]  perfect-reconstruction synthesis window: awin / overlap-added(awin^2) -- no division in the overlap-add
 *   the signal is padded with 1024 - hop ZEROS on both sides (no reflection): T frames cover L <= (T + 1) * hop - 1024
 *   samples, frame t starts at sample t * hop - (1024 - hop).
 * Both tables are built on the host (deepvoice3_pytorch_amd/audio.py: lws_windows).  Phase reconstruction is Griffin-Lim
 * (north_star) on this framing, not lws's own run_lws iterations. */
int dv3_lws_stft_f32(const float* y, const float* awin, float* phasor, float* spec, float* mag_bct, int32_t B, int32_t T,
                     int32_t hop, int32_t L, void* stream);
int dv3_lws_istft_frames_f32(const float* mag, const float* phasor /* NULL: zero phase */, const float* swin, float* frames,
                             int32_t B, int32_t T, void* stream);
/* y [B][(T + 1) * hop - 1024] */
int dv3_lws_overlap_add_f32(const float* frames, float* y, int32_t B, int32_t T, int32_t hop, void* stream);
int dv3_lws_gl_project_f32(const float* y, const float* mag, const float* awin, const float* swin, float* frames, int32_t B,
                           int32_t T, int32_t hop, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DV3HIP_H */
