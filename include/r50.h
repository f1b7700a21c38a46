/*
 * r50.h — C ABI of libr50hip.so: MI355X (gfx950) ResNet-50 feature extraction.
 *
 * The reference (ferreiraluisa/implementation-phd-lab-vision) has no FFI/plugin registry; its
 * boundary for this path is a Python call on an nn.Module-like object:
 *
 *     backbone = nn.Sequential(*list(resnet50(...).children())[:-1]).to(device).eval()
 *                                          -- src/preprocess_resnet_features.py:207-209
 *     feats = backbone(x).flatten(1)       -- src/preprocess_resnet_features.py:242,296
 *
 * Each entry point below cites the reference interface it replaces.  Plain pointers and sizes
 * only: no torch types, no exceptions across the boundary.  Every function returns 0 on success
 * or a negative r50_status; r50_last_error() gives the text.  One handle per (device, stream);
 * a handle is not re-entrant.  All device pointers are owned by the caller (e.g. PyTorch
 * allocations); the handle owns its packed weights and activation workspace.
 *
 * Layouts:  frames   fp32 NCHW (n,3,224,224), ImageNet-normalised, contiguous
 *                    (what src/dataset.py:242-245,429 produces and :295 reshapes)
 *           features fp32 (n,2048) row-major (= backbone(x).flatten(1), :296)
 *           internal activations bf16 NHWC.
 */
#ifndef R50_H_
#define R50_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct r50_handle r50_handle;

typedef enum r50_status {
    R50_OK = 0,
    R50_ERR_INVALID = -1,   /* bad argument (null pointer, n out of range, unknown name ...) */
    R50_ERR_HIP = -2,       /* a HIP runtime call failed */
    R50_ERR_STATE = -3,     /* e.g. forward before weights were loaded */
    R50_ERR_NOMEM = -4
} r50_status;

typedef enum r50_precision {
    R50_PREC_BF16 = 1,      /* bf16 operands, fp32 MFMA accumulation (the reference's CUDA autocast
                               dtype, src/preprocess_resnet_features.py:290-294) */
    R50_PREC_FP32X = 2,     /* fp32-class accuracy on the bf16 matrix cores: every value travels as a bf16
                               (head, tail) pair, each conv is three bf16 MFMA products with fp32 accumulation
                               (the reference's CPU numerics, autocast disabled, :239-241; ~3x the bf16 cost) */
    R50_PREC_BF16W2 = 3,    /* bf16 activations, every bottleneck conv weight as a (head, tail) pair of bf16: two MFMA
                               products per conv, one fp32 accumulator, activation traffic as in bf16 mode.  The bf16
                               error of this network is dominated by WEIGHT rounding, so this is enough to bring the
                               features within 1e-3 (rel-L2) of the fp32 reference at about half the fp32x cost */
    R50_PREC_FP16 = 4,      /* IEEE half operands and activations, fp32 MFMA accumulation: the bf16 path with the other
                               16-bit format (same kernels, same traffic, same speed; conversions saturate at 65504).
                               11 significand bits instead of 8 put the features 3e-4 from the fp32 reference */
    R50_PREC_FP8 = 5        /* BASELINE configs[4]: stem + layer1 as in bf16 mode (their 64-channel convs cannot fill a
                               128-byte fp8 K row), layer2-4 with OCP e4m3 weights and activations on the K = 128 scaled fp8
                               MFMA (per-tensor scales, fp32 accumulation).  Needs r50_set_fp8_scales before the first
                               forward.  A throughput mode: the features are ~1e-1 (rel-L2) from the fp32 reference */
} r50_precision;

/* One host tensor handed to r50_load_weights: torchvision state-dict key + fp32 data. */
typedef struct r50_tensor_desc {
    const char* name;       /* e.g. "layer1.0.conv1.weight", "bn1.running_var" */
    const float* data;      /* host pointer, contiguous fp32 (conv weights OIHW) */
    int64_t numel;
} r50_tensor_desc;

/* Replaces: backbone construction + .to(device) (preprocess_resnet_features.py:207-209).
 * Allocates workspace for batches of up to max_batch frames on HIP device `device_id`. */
int r50_create(r50_handle** out, int device_id, int precision, int max_batch);

/* Replaces: resnet50(weights=IMAGENET1K_V2) state loading + .eval() (:207-209).
 * Takes every conv weight and BN weight/bias/running_mean/running_var by torchvision key name
 * (fc.* and num_batches_tracked are not needed), folds eval-mode BN (eps 1e-5) into conv
 * weight+bias in fp32, converts to bf16 and uploads in the kernels' packed layout.
 * The caller keeps ownership of the host buffers.  R50_ERR_STATE for a handle that takes part in r50_share_weights (a sharer, or an
 * owner that still has sharers): its buffers are read by other handles' launches -- load into a fresh handle instead. */
int r50_load_weights(r50_handle* h, const r50_tensor_desc* tensors, int n_tensors);

/* Instead of r50_load_weights: `h` (fresh from r50_create, same device and precision as `from`, bf16 or fp16) reads the folded / packed weight buffers
 * of `from` -- a second backbone copy for a second batch in flight (backbone.BackboneLanes) without a second copy of the 47 MB of weights.  `h` keeps
 * its own activation workspace, streams, options and profile.  Either handle may be destroyed first: the buffers go with the last one. */
int r50_share_weights(r50_handle* h, r50_handle* from);

/* Replaces: backbone(x).flatten(1) (:242,296).  Asynchronous on `stream` (a hipStream_t; NULL =
 * default stream).  x_nchw_f32_dev: (n,3,224,224) fp32 on the device; out_f32_dev: (n,2048).
 * n may exceed max_batch: the call then loops over chunks of max_batch frames. */
int r50_forward(r50_handle* h, const float* x_nchw_f32_dev, int n, float* out_f32_dev, void* stream);

/* Same path, one step further upstream: frames as the resized uint8 crops the reference
 * holds right before `frames.to(torch.float32) / 255.0` (src/dataset.py:141-150) and `Normalize(mean, std)`
 * (:242-245), i.e. uint8 NCHW (n,3,224,224).  The three fp32 operations ((u8/255) - mean[c]) / std[c] are done
 * in the stem kernel, bit-identical to the host path; the boundary moves 150,528 B per frame instead of 602,112. */
int r50_forward_u8(r50_handle* h, const uint8_t* x_nchw_u8_dev, int n, float* out_f32_dev, void* stream);

/* Debug hook for per-layer parity tests (no reference counterpart; equivalent to a forward hook
 * on the nn.Sequential).  Runs the network on x (n <= max_batch) and copies the named bf16 NHWC
 * activation into out_bf16_nhwc_dev.  Names: "stem", "pool", "layer{1..4}.{b}",
 * "layer{i}.{b}.t1", ".t2", ".ds".  dims_out receives {N,H,W,C}. */
int r50_forward_layer(r50_handle* h, const float* x_nchw_f32_dev, int n, const char* layer,
                      void* out_bf16_nhwc_dev, int64_t out_capacity_bytes, int64_t dims_out[4],
                      void* stream);

/* Options: "micro_batch" (frames per pass through the layer stack, 0 = whole batch),
 * "profile" (1 = bracket every kernel launch with HIP events, see r50_profile_*),
 * "streams" (1..4: split the batch over internal streams forked from / joined to the caller's; default 1),
 * "overlap_ds" (1 = downsample convs on a side stream; default 0), "fused_stem" (default 1),
 * "fuse_tail" (layer1 / layer2: conv3 + identity + ReLU + the next block's conv1 in one kernel; default 1),
 * "fuse_tail3" (layer3.1-.4: the same pair chained through LDS in one launch; default 1; needs "fuse_tail"),
 * "fuse_block1" (layer1: the same for the 56x56 body, bneck_block1_kernel; 1 = layer1.1, 2 = layer1.2 as well, 3 = layer1.0 too, with its
 * downsample conv computed in the kernel; default 3; needs "fuse_tail"; same bits),
 * "fuse_tail3_last" (layer3.5, whose next conv1 does not fit the chain: conv3 + identity + ReLU alone through the pipelined tail kernel; default 1;
 * same bits),
 * "fuse_cat_chain" (layer2.0: conv3 + downsample + ReLU as one two-source conv chained with layer2.1.conv1 in one launch,
 * bneck_catchain_kernel; default 1; same bits),
 * "sub_out" (layer1.2 stores only the even rows and columns of its output, the ones layer2.0's stride-2 downsample conv reads -- layer2.0.conv1
 * is computed in the same launch; default 1; needs "fuse_block1" >= 2 and "fuse_cat_chain"; same features; debug taps always see full tensors),
 * "fuse_block2" (layer2.1-.3: conv2 + conv3 + identity + ReLU [+ the next conv1] in one launch, t2 kept in LDS; default 1; needs
 * "fuse_tail"; same bits),
 * "fuse_fp8_handover" (R50_PREC_FP8: layer1's output quantised to e4m3 in layer1.2.conv3's epilogue instead of in a pass of its
 * own; default 1; same bits either way),
 * "tile" (force an igemm tile id for every conv, 0 = tuned table; also turns "fuse_tail" off),
 * measurement knobs, all results bit-identical: "inplace_out" (1 = plain-identity blocks write their output over their input;
 * default 0), and PROCESS-WIDE ones (they apply to every handle of the process): "cu_cap" (workgroups a persistent launch may use,
 * 0 = every CU; for pipelines that share the chip), "tail3_bp" (real pixels per tile of the chained layer3 tail, 0 = automatic), "use_g8" (the
 * eight-phase GEMM tiles of gemm8p_kernel for the streaming 1x1 convs: 0 = never, 1 = on the shapes where they measured faster, the default,
 * 2 / 3 = wherever the shape fits, 256 / 224 pixels per tile), "use_s2" (0 = generic tiles for the stride-2 3x3 shapes) and "stem_strip"
 * (pooled-row pairs per strip of the fused stem kernel, 0 = chosen from the batch). */
int r50_set_option(r50_handle* h, const char* key, int64_t value);
int r50_get_option(r50_handle* h, const char* key, int64_t* value);

/* Per-kernel timing from HIP events recorded on the launch stream while option "profile" is 1.
 * r50_profile_collect synchronises the recorded events and accumulates them; then
 * r50_profile_count / r50_profile_entry enumerate kernel classes ("igemm", "conv1", "maxpool",
 * "avgpool", "stem_pack") and then one entry per bottleneck conv (named by its torchvision key, e.g.
 * "layer3.4.conv2"; "conv1" stays in its class), with launch count, total ms, algorithmic flops and bytes. */
int r50_profile_reset(r50_handle* h);
int r50_profile_collect(r50_handle* h);
int r50_profile_count(r50_handle* h);
int r50_profile_entry(r50_handle* h, int i, const char** name, int64_t* launches, double* total_ms,
                      double* flops, double* bytes);

/* Debug hook: copy the folded+packed parameters of one conv back to the host (BN-fold parity
 * tests).  conv_key e.g. "layer2.0.downsample.0"; what: 0 = bf16 weights in (cout,k,k,cin) order
 * (stem: the kernel's [kh][row][8][4] image), 1 = fp32 folded bias. */
int r50_get_packed(r50_handle* h, const char* conv_key, int what, void* dst_host, int64_t capacity_bytes,
                   int64_t* bytes_out);

/* R50_PREC_FP8 only: per-tensor activation scales (real value = stored fp8 value x scale) of the fp8 part, in execution order:
 * scales[0] = layer1's output (the bf16 -> fp8 hand-over), then per bottleneck of layer2, layer3, layer4: conv1 output, conv2 output,
 * [downsample output, first block of a layer only], block output: R50_FP8_NUM_SCALES values (calibrated on the host, e.g. 1.25 x
 * absmax / 448 of the bf16 network's tensors: backbone.py).  Weight scales are chosen by r50_load_weights (absmax / 448 per conv). */
#define R50_FP8_NUM_SCALES 43
int r50_set_fp8_scales(r50_handle* h, const float* scales, int n);
const char* r50_last_error(r50_handle* h);   /* h may be NULL: last error of r50_create / r50_op_* */
void r50_destroy(r50_handle* h);             /* replaces: del backbone */
const char* r50_version(void);

/* ---- Op-level entry points (per-kernel parity tests; each mirrors one nn.Module of the
 * reference's Sequential, upstream torchvision models/resnet.py).  bf16 NHWC device buffers. ---- */

/* conv2d(k x k, stride, pad, bias=folded BN) [+ residual] [+ ReLU].  w_ohwi_bf16: (cout,k,k,cin)
 * bf16; bias fp32 (cout); residual/y: (n,ho,wo,cout) bf16.  cin % 64 == 0, cout % 64 == 0,
 * k in {1,3}.  tile: 0 = auto, else a tile-config id.  */
int r50_op_conv2d(const void* x_nhwc_bf16, int n, int h, int w, int cin, const void* w_ohwi_bf16,
                  const float* bias_f32, const void* residual_nhwc_bf16, void* y_nhwc_bf16,
                  int cout, int ksize, int stride, int pad, int relu, int tile, void* stream);
/* The same with IEEE-half tensors (R50_PREC_FP16's element type). */
int r50_op_conv2d_f16(const void* x_nhwc_f16, int n, int h, int w, int cin, const void* w_ohwi_f16,
                  const float* bias_f32, const void* residual_nhwc_f16, void* y_nhwc_f16,
                  int cout, int ksize, int stride, int pad, int relu, int tile, void* stream);

/* fp8 convolution (BASELINE configs[4]: CDNA4 fp8 MFMA), kernel level.  OCP e4m3 activations, weights and output, fp32 accumulation on
 * v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales).  x (n,h,w,cin) fp8, w (cout,k,k,cin) fp8, residual / y (n,ho,wo,cout) fp8;
 * with per-tensor scales sx, sw, sr, sy (real value = stored value x scale):
 *   y = fp8( act( acc * oscale + residual * rscale ) ),  acc = bias_scaled + sum x*w,
 *   bias_scaled = bias / (sx*sw) (device fp32), oscale = sx*sw/sy, rscale = sr/sy; conversion saturates at +-448.
 * cin % 128 == 0, cout % 64 == 0 (128 / 256 for the wider tiles), k in {1,3}.  tile: 0 = auto or a role-specialised tile id (64|...). */
int r50_op_conv2d_fp8(const void* x_nhwc_fp8, int n, int h, int w, int cin, const void* w_ohwi_fp8, const float* bias_scaled_f32,
                      const void* residual_nhwc_fp8, void* y_nhwc_fp8, int cout, int ksize, int stride, int pad, int relu,
                      float oscale, float rscale, int tile, void* stream);

/* 1x1 conv over TWO K sources, the form the library runs conv3 + downsample + add + ReLU of a stage's first bottleneck in
 * (torchvision Bottleneck.forward: `out = conv3(out); identity = downsample(x); out += identity; relu`): y = act([W1 | W2] .
 * [x1 ; x2 sampled at stride2] + bias).  x1 (n,h,w,c1) at the output resolution, x2 (n,h2,w2,c2) with (h2-1)/stride2+1 == h;
 * wcat (cout, c1+c2) K-major; c1, c2, cout multiples of 64.  tile: 0 = auto or a role-specialised tile id (64|...). et: 0 bf16, 1 fp16. */
int r50_op_conv1x1_cat(const void* x1, int n, int h, int w, int c1, const void* x2, int h2, int w2, int c2, int stride2,
                       const void* wcat, const float* bias_f32, void* y, int cout, int relu, int tile, int et, void* stream);

/* Stem: fp32 NCHW frames -> conv 7x7 s2 p3 (+folded bn1 bias) + ReLU -> (n,112,112,64) bf16 NHWC.
 * w_oihw_f32: (64,3,7,7) fp32 *already folded*, host pointer; bias_f32: device pointer (64).
 * scratch_dev: at least r50_stem_scratch_bytes(n) bytes of device memory. */
int64_t r50_stem_scratch_bytes(int n);
int r50_op_stem(const float* x_nchw_f32_dev, int n, const float* w_oihw_f32_host,
                const float* bias_f32_dev, void* scratch_dev, void* y_nhwc_bf16, void* stream);

/* MaxPool2d(3, stride 2, pad 1) on bf16 NHWC; c % 8 == 0. */
int r50_op_maxpool(const void* x_nhwc_bf16, int n, int h, int w, int c, void* y_nhwc_bf16, void* stream);

/* Bottleneck tail in one launch: conv3 1x1 (cmid -> 4*cmid) + bn3 + identity + ReLU -> out (m,4*cmid), and the
 * next block's conv1 1x1 (4*cmid -> c1) + bn1 + ReLU -> y1n (m,c1).  Replaces, for two consecutive torchvision
 * Bottleneck blocks, `out = relu(bn3(conv3(out)) + identity)` of the first and `out = relu(bn1(conv1(x)))` of
 * the second (upstream torchvision models/resnet.py Bottleneck.forward; the reference builds them at
 * src/preprocess_resnet_features.py:207).  Shapes: cmid = 64 (layer1; c1 in {64,128}) or cmid = 128 (layer2;
 * c1 = 128, no wd/bd).
 * wd/bd NULL: `identity` is the (m,4*cmid) identity tensor.  wd/bd given (cmid = 64, the stage's first block):
 * `identity` is the block INPUT (m,64) and the identity is `downsample(x)` = bf16(wd . x + bd), computed in
 * the kernel with the same rounding as a separate launch.  All tensors bf16 NHWC with m = n*h*w pixels,
 * weights folded (cout, cin) K-contiguous, biases fp32.  cmid = 64 reproduces the two separate launches bit
 * for bit; cmid = 128 sums the second conv's K in eight slices (fp32), i.e. within rounding of them; cmid = 256
 * (c1 = 256, layer3: the chained kernel, weights packed into fragment order per call by this hook) is bit for
 * bit again.  cmid = 256 with c1 = 0 and w1 = b1 = y1n = NULL: conv3 + identity + ReLU ALONE through the same pipelined kernel (the form the
 * stage's last block, layer3.5, runs in; bit for bit a 1x1 r50_op_conv2d with a residual).  (Environment R50_TAIL3_BP = 1..112: pixels per tile
 * of the cmid = 256 kernel; a test knob.) */
int r50_op_bneck_tail(const void* y2_bf16, int64_t m, int cmid, const void* w3_bf16, const float* b3,
                      const void* identity_bf16, const void* wd_bf16, const float* bd, void* out_bf16, const void* w1_bf16,
                      int c1, const float* b1, void* y1n_bf16, void* stream);

/* A whole bottleneck BODY of layer2 (blocks .1-.3, 28x28, 128 mid channels) in one launch: t2 = relu(conv2_3x3(t1) + b2) stays in LDS,
 * out = relu(w3 . t2 + b3 + identity) and, when w1 / b1 / y1n are given, y1n = relu(w1 . out + b1) (the next block's conv1) --
 * `Bottleneck.forward` of torchvision's ResNet-50 from its second conv on (src/preprocess_resnet_features.py:296 calls it through
 * nn.Sequential).  t1 (n,28,28,128), identity / out (n,28,28,512), y1n (n,28,28,128) bf16 NHWC; w2 (128,3,3,128), w3 (512,128),
 * w1 (128,512) folded bf16, K contiguous; biases fp32.  Bit for bit what r50_op_conv2d (3x3, input-resident tile) followed by two
 * r50_op_conv2d 1x1 launches give.  w1 = b1 = y1n = NULL: no next conv1 (the stage's last block). */
int r50_op_bneck_block2(const void* t1_bf16, int n, const void* w2_bf16, const float* b2, const void* w3_bf16, const float* b3,
                        const void* identity_bf16, void* out_bf16, const void* w1_bf16, const float* b1, void* y1n_bf16, void* stream);

/* layer1.0, the stage's first bottleneck, from its second conv on in one launch: as r50_op_bneck_block1 with c1 = 64, but the identity is the
 * DOWNSAMPLE conv of the block input computed in the kernel -- identity = bf16(wd . x + bd), rounded exactly as a separate 1x1 launch stores
 * it (torchvision `Bottleneck.forward`: `identity = self.downsample(x)`).  x (n,56,56,64) the block input, wd (256,64), bd (256).  Bit for bit
 * what the resident-weights 3x3 launch followed by r50_op_bneck_tail (cmid 64, wd / bd given) give. */
int r50_op_bneck_block1_ds(const void* t1_bf16, int n, const void* w2_bf16, const float* b2, const void* w3_bf16, const float* b3,
                           const void* x_bf16, const void* wd_bf16, const float* bd, void* out_bf16, const void* w1_bf16, const float* b1,
                           void* y1n_bf16, void* stream);

/* The TRANSITION tail of layer2.0 chained with layer2.1.conv1 in one launch: out = relu([W3 | Wd] . [t2 ; x at stride 2] + (b3 + bd)) -- conv3,
 * the downsample conv, the add and the ReLU of torchvision's first Bottleneck of a stage (`out = relu(bn3(conv3(out)) + downsample(x))`,
 * src/preprocess_resnet_features.py:296 through nn.Sequential) as one 1x1 conv over two K sources, as r50_op_conv1x1_cat computes it -- and
 * y1n = relu(w1 . out + b1), the next block's conv1.  t2 (n,ow,ow,128), x (n,2ow,2ow,256), out (n,ow,ow,512), y1n (n,ow,ow,128) bf16 NHWC;
 * wcat (512, 128 + 256) = [W3 | Wd], w1 (128,512) folded bf16, K contiguous; bcat = b3 + bd, b1 fp32.  ow = 28.  Bit for bit what
 * r50_op_conv1x1_cat followed by a 1x1 r50_op_conv2d give. */
int r50_op_bneck_cat_chain(const void* t2_bf16, const void* x_bf16, int n, int ow, const void* wcat_bf16, const float* bcat, void* out_bf16,
                           const void* w1_bf16, const float* b1, void* y1n_bf16, void* stream);

/* The same for layer1 (blocks .1 / .2, 56x56, 64 mid channels): t1 (n,56,56,64), identity / out (n,56,56,256), w2 (64,3,3,64), w3 (256,64),
 * w1 (c1,256) with c1 = 64 (the next layer1 block's conv1) or 128 (layer2.0.conv1), y1n (n,56,56,c1).  Bit for bit what the
 * resident-weights 3x3 launch followed by two r50_op_conv2d 1x1 launches give. */
int r50_op_bneck_block1(const void* t1_bf16, int n, const void* w2_bf16, const float* b2, const void* w3_bf16, const float* b3,
                        const void* identity_bf16, void* out_bf16, const void* w1_bf16, int c1, const float* b1, void* y1n_bf16, void* stream);

/* Frame producer, the step before the path: crop box + bilinear resize of a decoded clip on the
 * device.  Replaces `_crop_and_resize_video_uint8` (src/dataset.py:141-149) up to, not including, the `/255`:
 * frames (t,h,w,3) uint8 HWC as the video decoder returns them -> `frames[:, top:top+hh, left:left+ww]` ->
 * resize to (out_size,out_size), bilinear, no antialias -> out (t,3,out_size,out_size) uint8 NCHW, the input of
 * r50_forward_u8.  mode R50_RESIZE_FLOAT: the arithmetic of `torchvision.transforms.functional.resize` (the v1 API
 * the reference imports, :13): uint8 -> fp32, ATen bilinear (align_corners=False), round half to even, uint8.
 * mode R50_RESIZE_FIXED: ATen's native uint8 kernel (what the v2 API runs on an AVX2 CPU): two-pass int16 fixed
 * point with a uint8 intermediate, bit-exact.  The box comes from `_compute_square_crop_from_2d` (:75-104; host
 * code, mirrored in frames.py) and must lie inside the frame; out_size % 4 == 0.  Device pointers; asynchronous on
 * `stream` (indices and weights are computed in the kernel).
 * flags: the two augmentation variants that are index permutations, applied to the output exactly as
 * the reference applies them to the resized clip: R50_AUG_HFLIP = `torch.flip(video, dims=[-1])` (`_aug_hflip`, :158-166),
 * R50_AUG_TREV = `torch.flip(video, dims=[0])` (`_aug_temporal_reverse`, :199-207); joints / intrinsics are host code. */
#define R50_RESIZE_FLOAT 0
#define R50_RESIZE_FIXED 1
#define R50_AUG_HFLIP 1
#define R50_AUG_TREV 2
int r50_op_crop_resize_u8(const void* frames_thwc_u8, int t, int h, int w, int top, int left, int hh, int ww,
                          void* out_tchw_u8, int out_size, int mode, int flags, void* stream);

/* ColorJitter augmentation variant: `_aug_color_jitter` (src/dataset.py:188-197) = torchvision.transforms.v2
 * ColorJitter(brightness=0.3, contrast=0.3, saturation=0.2, hue=0.05) on the float clip in [0,1], followed (normalize != 0) by
 * `frame_tf` = Normalize(ImageNet mean, std) (:242-245).  frames_u8: (t,3,hw) uint8 resized crops on the device (the output of
 * r50_op_crop_resize_u8); order4: the sampled permutation of {0 brightness, 1 contrast, 2 saturation, 3 hue} (HOST pointer); the
 * four factors as sampled (host code: frames.sample_color_jitter_params); out_f32: (t,3,hw) fp32, what r50_forward takes;
 * scratch_means: t floats of device memory (per-frame grayscale mean of adjust_contrast).  Asynchronous on `stream`. */
int r50_op_color_jitter_u8(const void* frames_u8, int t, int hw, const int* order4_host, float brightness, float contrast,
                           float saturation, float hue, int normalize, float* out_f32, float* scratch_means, void* stream);

/* Lifting head, forward (first step): the non-GEMM pieces of `PHDFor3DJoints.forward` (src/model.py:146-178);
 * its Linear layers and causal conv1d's run on r50_op_conv2d / r50_op_conv2d_f16 as 1x1 convolutions over the b*t rows.
 * et: 0 = bf16, 1 = fp16.  All device pointers.
 *  r50_op_cast_rows: fp32 (rows,c) -> element (rows,cpad), zero columns beyond c (`feats` before `input_proj`, :155).
 *  r50_op_concat_pad: [phi (rows,d) element | y (rows,ny) fp32 | zeros] -> (rows,dp) element = `torch.cat([phi, y], -1)` of the
 *    iterative regressor (:113) padded to the GEMM's K granularity.
 *  r50_op_gn_relu_causal3: x (b,t,c) element -> GroupNorm(groups, eps) + ReLU (`ResidualBlock`, :47-55) -> the input rows of
 *    the following `CausalConv1d` (kernel 3, replicate left padding, :20-35): out (b*t, 3c), row (b,t) =
 *    [y(t-2) | y(t-1) | y(t)] with indices clamped at 0.
 *  r50_op_add_rows: y (rows,ny) fp32 += dy (rows,dp) element, first ny columns (`y = y + dy`, :114-115). */
int r50_op_cast_rows(const float* src_f32, int64_t rows, int c, void* dst, int cpad, int et, void* stream);
int r50_op_concat_pad(const void* phi, int d, const float* y_f32, int ny, int64_t rows, void* dst, int dp, int et, void* stream);
int r50_op_add_rows(float* y_f32, int ny, const void* dy, int dp, int64_t rows, int et, void* stream);
int r50_op_gn_relu_causal3(const void* x, int b, int t, int c, int groups, const float* gamma, const float* beta, float eps,
                           void* out, int et, void* stream);

/* Lifting head, backward + optimizer: the training step of `train()` (src/train.py:137-176: fp16 autocast forward, `l3d` MSE
 * loss, GradScaler, AdamW).  Every matrix product of the backward pass (dX = dY W, dW = dY^T X) is an r50_op_conv2d(_f16) launch
 * on operands transposed by r50_op_transpose16; these are the pieces around them.  16-bit tensors are `et` elements (0 bf16, 1 fp16).
 *  r50_op_transpose16: src (rows,cols) -> dst (cols,ld), dst[c][r] = src[r][c]; ld >= rows, the padding columns are left alone.
 *  r50_op_mask_scale: x *= mask * scale in place (`nn.Dropout`, src/model.py:44,98); mask one byte per element.
 *  r50_op_relu_bwd: dy = dy * scale * (act > 0) in place (backward of ReLU, or of ReLU + dropout with act = the dropped-out tensor).
 *  r50_op_colsum / _f32: out (cols) fp32 [+]= scale * column sums of x (rows,ld)[:, :cols] (bias gradients; rows summed in order).
 *  r50_op_grad_accum: dst (n) fp32 [+]= scale * src (n) 16-bit (a weight gradient into the flat fp32 gradient buffer, unscaled).
 *  r50_op_mse_loss_grad: loss2[0] = mean((y-gt)^2) (:161), loss2[1] = MPJPE (:42-45); dy = 2 (y-gt) / n * loss_scale; n = B*T*J*3.
 *  r50_op_gn_relu_causal3_bwd: backward of r50_op_gn_relu_causal3: dr (b*t,3c) -> dx (b,t,c) [+ add], per-sample parameter
 *    gradient parts dgamma_part / dbeta_part (b,c) fp32 (sum over b with r50_op_colsum_f32).
 *  r50_op_check_finite: found[0] |= any non-finite in g (GradScaler's inf check, :172-174).
 *  r50_op_check_overflow16: found[0] |= any inf / nan in 16-bit x -- or, for fp16, any element at +-65504: the fp32 -> fp16 conversion
 *    of this library saturates instead of producing inf, so that is what an overflowed gradient looks like.
 *  r50_op_adamw: `torch.optim.AdamW` (:389) over flat fp32 p / m / v / g, step >= 1; skipped when found_inf[0] != 0; p16 = the
 *    refreshed 16-bit copy of p the GEMMs read. */
int r50_op_transpose16(const void* src, int rows, int cols, void* dst, int ld, void* stream);
int r50_op_mask_scale(void* x, const void* mask_u8, float scale, int64_t n, int et, void* stream);
int r50_op_relu_bwd(void* dy, const void* act, float scale, int64_t n, int et, void* stream);
int r50_op_colsum(const void* x, int64_t rows, int cols, int ld, float scale, float* out_f32, int accumulate, int et, void* stream);
int r50_op_colsum_f32(const float* x, int64_t rows, int cols, float scale, float* out_f32, int accumulate, void* stream);
int r50_op_grad_accum(const void* src, float scale, float* dst_f32, int64_t n, int accumulate, int et, void* stream);
int r50_op_mse_loss_grad(const float* y, const float* gt, int64_t n, float loss_scale, float* dy, float* loss2, void* stream);
int r50_op_gn_relu_causal3_bwd(const void* dr, const void* x, int b, int t, int c, int groups, const float* gamma, const float* beta,
                               float eps, const void* add, void* dx, float* dgamma_part, float* dbeta_part, int et, void* stream);
int r50_op_check_finite(const float* g, int64_t n, int* found, void* stream);
int r50_op_check_overflow16(const void* x, int64_t n, int* found, int et, void* stream);
int r50_op_adamw(float* p, float* m, float* v, const float* g, void* p16, int64_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step, const int* found_inf, int et, void* stream);

/* AdaptiveAvgPool2d((1,1)) + flatten(1): (n,hw,c) bf16 -> (n,c) fp32; c % 8 == 0. */
int r50_op_avgpool(const void* x_nhwc_bf16, int n, int hw, int c, float* y_f32, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* R50_H_ */
