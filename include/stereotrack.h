/*
 * stereotrack.h — C ABI of libstereotrack_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the per-frame dense compute path of
 * Superjie13/StereoTracking.  The reference has NO FFI of its own (pure Python
 * on third-party wheels, reference setup.py:223 `ext_modules=[]`), so every
 * entry point below cites the reference *Python* interface it replaces; the
 * ctypes binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, a negative ST_ERR_* otherwise;
 *     st_last_error() returns a thread-local message.  No exceptions cross
 *     the ABI.
 *   - all pointers named *_dev are device pointers owned by the CALLER
 *     (PyTorch); the library owns only packed weights (freed by *_destroy).
 *   - all work is enqueued on the caller's stream (a hipStream_t passed as
 *     void*); no call synchronises the device or allocates on the hot path.
 *   - activations are fp32.  Internal activation layout is NHWC
 *     (pixel-major, channel-contiguous); the boundary tensors keep the
 *     reference's NCHW layout.
 */
#ifndef STEREOTRACK_H_
#define STEREOTRACK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ST_VERSION 430

enum {
  ST_OK = 0,
  ST_ERR_INVALID = -1,   /* bad argument / shape mismatch            */
  ST_ERR_HIP = -2,       /* a HIP runtime call failed                */
  ST_ERR_STATE = -3,     /* call order violated (e.g. not finalized) */
  ST_ERR_WORKSPACE = -4, /* caller workspace too small               */
  ST_ERR_NOTFOUND = -5   /* unknown parameter name                   */
};

typedef void* st_stream_t; /* hipStream_t */

int st_version(void);
const char* st_last_error(void);

/* ------------------------------------------------------------------------
 * 1. Convolution primitive (exact-fp32 MFMA implicit GEMM, NHWC).
 *    Replaces one mmcv ConvModule = Conv2d(bias=False) + BatchNorm2d(eps) +
 *    SiLU as built in reference
 *    mmtrack/models/backbones/csp_darknet_disparity_v1.py:104-153
 *    (BN already folded into wgt/bias by the caller or by st_detector_*).
 * ---------------------------------------------------------------------- */
typedef struct StConvDesc {
  /* input: NHWC view, pixel (n,y,x) at in_dev[((n*Hi+y)*Wi+x)*in_ld + in_off + c] */
  const float* in_dev;
  int N, Hi, Wi, Cin, in_ld, in_off;
  /* weights [CoutPad][Kpad] row-major, K index = (kh*KW+kw)*Cin + ci,
   * Kpad = roundup(KH*KW*Cin, 32), CoutPad = roundup(Cout, 32), zero padded;
   * bias [CoutPad] */
  const float* wgt_dev;
  const float* bias_dev;
  int Cout, KH, KW, stride, pad;
  /* outputs: channels [0,split) -> out1, [split,Cout) -> out2 (NULL if unused) */
  float* out1_dev;
  int out1_ld, out1_off, split;
  float* out2_dev;
  int out2_ld, out2_off;
  /* optional: every channel is ALSO stored nearest-x2 upsampled (2Ho x 2Wo) */
  float* up_dev;
  int up_ld, up_off;
  /* optional residual: out = (act(conv+bias) + res) * post_scale */
  const float* res_dev;
  int res_ld, res_off;
  float post_scale;
  int act; /* 0 = identity, 1 = SiLU */
  /* optional: the same weights in Winograd F(2x2,3x3) form (st_wino_pack_weights of wgt, fragment order); enables
   * kernel instances 43 / 44 for 3x3 / stride-1 / pad-1 layers with Cin % 4 == 0 (>= 16; a Cin that is not a multiple
   * of 32 runs a short last K-chunk) and Cout a multiple of 32 or 33..64 (one zero-padded block of 64).  NULL = not
   * available. */
  const float* wgt_wino_dev;
} StConvDesc;

int st_conv2d_nhwc(const StConvDesc* d, st_stream_t stream);
/* The same convolution with the kernel instance chosen by the caller instead of the library's heuristic:
 * variant 0..21 = tile instances of the implicit-GEMM kernel (st_conv_variant_name), 41 = streaming 1x1 kernel,
 * 42 = direct 3x3 kernel, 43 / 44 = Winograd F(2x2,3x3) kernel with 64- / 32-cout workgroups (need wgt_wino_dev),
 * 46 = 1x1 kernel with the weight matrix resident in LDS and register-fed pixels (Cin -> Cout one of 64->64, 128->64,
 * 128->128, 256->128), -1 = heuristic.  Returns ST_ERR_INVALID when the instance cannot run this layer (tile
 * does not divide the padded Cout, ...), so callers can autotune per layer by timing the valid ones — which is
 * what st_detector_autotune does internally and StereoCostVolume.autotune does for the aggregation convs.
 * Results are the same convolution for every valid variant (fp32 rounding differs with the summation order). */
int st_conv2d_nhwc_variant(const StConvDesc* d, st_stream_t stream, int variant);
/* Winograd form of a 3x3 weight tensor.  packed_wgt_host: the [CoutPad][Kpad] matrix st_conv_pack_weights produced
 * (host memory); out_host: st_wino_packed_floats(Cout, Cin) floats, to be uploaded and passed as wgt_wino_dev.
 * U = G g G^T is evaluated in fp64 on the folded fp32 weights and rounded once. */
size_t st_wino_packed_floats(int Cout, int Cin);
int st_wino_pack_weights(const float* packed_wgt_host, int Cout, int Cin, float* out_host);

/* Fused pair of 1x1 convolutions for the narrow high-resolution CSP layers: `b` (Cin = 32, Cout <= 32) consumes
 * output channels [0, 32) of `a` (Cin 32 or 64, 32 < Cout <= 64, no residual) - the CSPLayer main_conv ->
 * DarknetBottleneck conv1 pair (mmdet CSPLayer, built at csp_darknet_disparity_v1.py:113-153).  a's outputs are
 * written as usual; b's in_dev / in_ld / in_off are ignored (its input never leaves the registers).  The same pair
 * one stage deeper (a: 128 or 256 -> 64 | 64 split store, b: 64 -> 64) runs on the LDS-resident kernel (variant 46). */
int st_conv1x1_chain(const StConvDesc* a, const StConvDesc* b, st_stream_t stream);
/* Fused head of a stage-1 CSP branch: `a` = 3x3 / stride-2 / pad-1 ConvModule 32 -> 64 (its own output tensor is never
 * written: a->out1_dev is ignored), `ms` = CSPLayer main_conv | short_conv on a's output (1x1, 64 -> 32 | 32, split
 * store to ms->out1 / ms->out2), `c1` = DarknetBottleneck conv1 on the main half (1x1, 32 -> 32, to c1->out1).  One
 * launch instead of three; the 64-channel stride-2 tensor stays in MFMA accumulators.  Replaces the module sequence
 * csp_darknet_disparity_v1.py:113-153 builds for `stage1` / `disp_stage1` (ConvModule(c, 2c, 3, stride=2) followed by
 * mmdet CSPLayer) as run at :176-183.  All three: SiLU, no residual / upsample store / post_scale.  frag_*_dev =
 * st_front_pack_frags of the ms / c1 packed weight matrices (device copies).  ST_ERR_INVALID when the shapes differ. */
size_t st_front_frag_floats(int Cout, int Cin);
int st_front_pack_frags(const float* packed_wgt_host, int Cout, int Cin, float* out_host);
int st_conv3x3s2_csp_front(const StConvDesc* a, const StConvDesc* ms, const StConvDesc* c1, const float* frag_ms_dev,
                           const float* frag_c1_dev, st_stream_t stream);
/* Fused TAIL of a stage-1 CSP branch (round 6, tile variant 56): `conv2` = DarknetBottleneck conv2 (3x3 / stride 1 /
 * pad 1, 32 -> 32, SiLU, with the identity in conv2->res) whose output tensor conv2->out1 (the first 32 channels of the
 * CSP concat) is NEVER written, `fin` = CSPLayer final_conv (1x1, 64 -> 64, SiLU) reading that concat (fin->in_dev ==
 * conv2->out1_dev, same ld / offset; its channels [32, 64) = the `short` half must already be in memory), with an
 * optional residual on `fin` = the two-branch average (v + other) * post_scale of
 * csp_darknet_disparity_v1.py:155-184.  One PERSISTENT launch instead of two: Winograd F(2x2,3x3) with the transformed
 * 32 x 32 weights resident in registers, conv2's output handed to the 1x1 GEMM through LDS.  Replaces the module
 * sequence mmdet CSPLayer builds behind its first bottleneck (reference csp_darknet_disparity_v1.py:145-153, run at
 * :176-184).  conv2 needs wgt_wino_dev; frag_fin_dev = st_csp_tail_pack_frags of fin's packed weight matrix (device
 * copy, st_csp_tail_frag_floats() floats).  ST_ERR_INVALID when the shapes differ. */
size_t st_csp_tail_frag_floats(void);
int st_csp_tail_pack_frags(const float* packed_wgt_host, float* out_host);
int st_conv3x3_csp_tail(const StConvDesc* conv2, const StConvDesc* fin, const float* frag_fin_dev, st_stream_t stream);
/* Pack one Conv2d weight [Cout][Cin][KH][KW] (+ optional BN, folded in fp64)
 * into the kernel layout above.  Host function; out buffers are host memory
 * of st_conv_packed_floats(...) / roundup(Cout,32) floats. */
size_t st_conv_packed_floats(int Cout, int Cin, int KH, int KW);
int st_conv_pack_weights(const float* w, const float* conv_bias, /* may be NULL */
                         const float* bn_gamma, const float* bn_beta,
                         const float* bn_mean, const float* bn_var, /* NULL = no BN */
                         double bn_eps, int Cout, int Cin, int KH, int KW,
                         float* wgt_out, float* bias_out);

/* ------------------------------------------------------------------------
 * 2. Input packing: NCHW fp32 image -> Focus (space-to-depth) NHWC 12-ch.
 *    Replaces mmdet Focus slicing, reference
 *    csp_darknet_disparity_v1.py:104-111 (order TL,BL,TR,BR).
 * ---------------------------------------------------------------------- */
int st_focus_pack(const float* img_nchw_dev, int N, int C, int H, int W,
                  float* out_nhwc_dev, st_stream_t stream);

/* Fused Focus + stem ConvModule: Focus(x) followed by the 3x3/s1/p1 ConvModule over its 12 channels is a
 * 6x6/s2/p2 convolution of the planar 3-channel image, computed here in one kernel (no NHWC 12-channel
 * intermediate).  Replaces csp_darknet_disparity_v1.py:104-111 (`Focus(3, c, kernel_size=3)`, both the RGB
 * `stem` and the `disp_stem`) as run at :176-179.  Cout <= 64; H even, W a multiple of 4, image 16-byte aligned.
 *   w            stem conv weight [Cout][12][3][3], channels in Focus order (TL,BL,TR,BR groups of 3)
 *   wgt_out      st_stem_packed_floats(Cout) floats, bias_out round_up(Cout,32) floats (host buffers)
 *   out_nhwc_dev [N][H/2][W/2][out_ld], channels written at [out_off, out_off + Cout)          */
size_t st_stem_packed_floats(int Cout);
/* used_planes: 3 = generic image; 1 = the caller guarantees that the three planes are identical (disp_postp is
 * a 3-channel repeat of one map, loading_disparity.py:85-86): the per-tap weights of the three planes are summed
 * (fp64, rounded once) and only plane 0 is read: K = 36 instead of 108. */
int st_stem_pack_weights(const float* w, const float* conv_bias, /* may be NULL */
                         const float* bn_gamma, const float* bn_beta,
                         const float* bn_mean, const float* bn_var, /* NULL = no BN */
                         double bn_eps, int Cout, int used_planes, float* wgt_out, float* bias_out);
int st_stem_focus_conv(const float* img_nchw_dev, int N, int H, int W, int used_planes, const float* wgt_dev,
                       const float* bias_dev, int Cout, float* out_nhwc_dev, int out_ld, int out_off,
                       int act /* 1 = SiLU */, st_stream_t stream);
/* The same stem reading RAW frames: N separate uint8 [3][h][w] device frames (HOST array of N <= 32 device pointers,
 * 4-byte aligned, w % 4 == 0), converted to fp32 and padded to H x W with pad_value (integral, 0..255) while the input
 * windows are staged - TrackDataPreprocessor_Disparity_V1's cast + stack_batch pad (reference
 * data_preprocessor_disparity_v1.py:38-51, utils/misc.py:13-64) fused into the stem; same values as st_pack_raw_frames
 * followed by st_stem_focus_conv. */
int st_stem_focus_conv_u8(const unsigned char* const* frames_u8_dev_ptrs_host, int N, int h, int w, int H, int W,
                          float pad_value, const float* wgt_dev, const float* bias_dev, int Cout, float* out_dev,
                          int out_ld, int out_off, int act, st_stream_t stream);

/* Raw input packing: uint8 image (N,3,h,w) -> fp32 (N,3,H,W) padded with img_pad;
 * uint16 disparity PNG codes (N,h,w) -> disp_postp fp32 px = code/16 (65535 -> 0) x3 channels padded with
 * 0, and disp_mask (N,1,H,W) = code < 65535.  Replaces LoadDisparityFromFile._post_processing_v2
 * (reference datasets/transforms/loading_disparity.py:82-86,129-134), Pad_Disparity
 * (transforms_disparity.py:234-249) and the preprocessor's cast + pad
 * (data_preprocessor_disparity_v1.py:38-51) for frames uploaded raw.  Any of the two halves may be NULL. */
int st_pack_raw_inputs(const unsigned char* img_u8_dev, const unsigned short* disp_u16_dev, int N, int h, int w,
                       int H, int W, float img_pad, float* img_out_dev, float* disp_postp_out_dev,
                       float* disp_mask_out_dev, st_stream_t stream);
/* The image half of st_pack_raw_inputs for frames in SEPARATE allocations (one contiguous (3, h, w) uint8 tensor per
 * frame, as a dataloader hands them over - reference formatting_disparity.py:139-338 packs one frame per sample):
 * frames_u8_dev_ptrs_host = HOST array of N device pointers (N <= 32; they travel in the kernel arguments), w and W
 * multiples of 4.  No concatenated staging copy. */
int st_pack_raw_frames(const unsigned char* const* frames_u8_dev_ptrs_host, int N, int h, int w, int H, int W,
                       float img_pad, float* img_out_dev, st_stream_t stream);
/* Resize_Disparity with a non-identity scale (reference datasets/transforms/transforms_disparity.py:23-137: the image
 * through mmcv.imrescale / imresize = cv2.resize INTER_LINEAR, disp_postp / disp_mask / depth_postp through
 * INTER_NEAREST, :52-112).  P planes of h x w elements -> h2 x w2; interleaved = 0: planar [P][h][w], 1: [h][w][P].
 * bilinear = 1 (elem_bytes 1 only): OpenCV's 8-bit INTER_LINEAR restated (11-bit fixed-point taps, the exact 2 x 2
 * decimation as a box mean; cv2 is un-vendored: published algorithm, parity unpinned); bilinear = 0: INTER_NEAREST for
 * 1- / 2- / 4-byte elements (uint8 masks, uint16 PNG codes, fp32 maps: the sampling commutes with code / 16). */
int st_resize_planes(const void* in_dev, int P, int h, int w, int interleaved, void* out_dev, int h2, int w2,
                     int elem_bytes, int bilinear, st_stream_t stream);

/* SPP: out[..., 0:C]=x, [C:2C]=maxpool5, [2C:3C]=maxpool9, [3C:4C]=maxpool13
 * (stride 1, same pad, -inf padding).  x may alias out channels [0,C).
 * Replaces mmyolo SPPFBottleneck pooling, csp_darknet_disparity_v1.py:137-144 */
int st_spp_pool(const float* x_dev, int x_ld, int x_off, int N, int H, int W, int C,
                float* out_dev, int out_ld, int out_off, st_stream_t stream);

/* ------------------------------------------------------------------------
 * 3. Two-branch YOLOX detector (backbone + PAFPN + decoupled head).
 *    Replaces YOLODetector_Disparity_V1._forward, reference
 *    mmtrack/models/detectors/yolo_detector_disparity_v1.py:127-142
 *    (extract_feat :77-90 -> backbone forward
 *    csp_darknet_disparity_v1.py:155-206 -> mmyolo YOLOXPAFPN ->
 *    YOLOXHeadModule.forward).
 * ---------------------------------------------------------------------- */
typedef struct StDetector StDetector;

typedef struct StDetectorConfig {
  int struct_size;      /* = sizeof(StDetectorConfig) */
  float widen_factor;   /* 0.5 for YOLOX-s */
  float deepen_factor;  /* 0.33 */
  int num_classes;      /* 1 */
  int batch;            /* N frames per forward */
  int height, width;    /* padded input size, multiples of 32 */
  double bn_eps;        /* 1e-3 */
  int with_right_branch; /* also build stem+stage1 features for the right image (stereo module) */
  int disp_planes_identical; /* 1 = the caller guarantees disp_postp is a 3-channel repeat of ONE map (what the
                              * reference loader yields, loading_disparity.py:85-86, and what
                              * st_disp_upsample_pack writes): the disparity stem then reads plane 0 only with
                              * plane-summed weights (K = 36 instead of 108).  0 = generic 3-plane input. */
  int rgb_only;         /* 1 = the single-branch detector of the reference's second stereo config
                         * (configs/stereo_tracking/ocsort/yolox_s_mmyolo_mot_airdrone.py:40-42: backbone
                         * `mmtrack.CSPDarknet`, mmtrack/models/backbones/csp_darknet.py:8-13 - forward reads x['img']
                         * only): no disp_stem / disp_stage1 parameters or launches, no branch average; stage2 consumes
                         * the RGB stage-1 features.  The disparity input of the forward calls may then be NULL (the
                         * MOT shell still consumes the disparity for the per-box depth, ocsort_disparity.py:82-83). */
} StDetectorConfig;

int st_detector_create(const StDetectorConfig* cfg, StDetector** out);
int st_detector_destroy(StDetector* det);

/* Parameter table, named exactly like the reference state_dict
 * ("backbone.stem.conv.conv.weight", "backbone.disp_stage1.0.bn.running_var",
 *  "neck.reduce_layers.2.conv.weight", "bbox_head.head_module.multi_level_conv_cls.0.bias" ...;
 *  SURVEY.md §5 checkpoint row). */
int st_detector_num_params(const StDetector* det);
int st_detector_param_info(const StDetector* det, int idx, char* name, int name_cap,
                           int64_t shape[4], int* ndim);
int st_detector_set_param(StDetector* det, const char* name, const float* host, int64_t numel);
/* fold BN (fp64), repack, upload.  Needs a current HIP device. */
int st_detector_finalize(StDetector* det);

size_t st_detector_workspace_bytes(const StDetector* det);
/* head output layout: for level l (stride 8,16,32): float[N][H_l*W_l][row] rows of
 * [cls logits (num_classes) | reg x,y,w,h | obj logit | unused up to st_head_row_floats(num_classes): 8 for 1..3 classes];
 * levels concatenated.  st_detector_head_floats = total float count. */
size_t st_detector_head_floats(const StDetector* det);
int st_detector_num_levels(const StDetector* det);
int st_detector_level_info(const StDetector* det, int level, int* h, int* w, int* stride,
                           size_t* float_offset);
/* img/disp: NCHW fp32 [N][3][H][W] device pointers (the tensors
 * TrackDataPreprocessor_Disparity_V1 produces, reference
 * data_preprocessor_disparity_v1.py:21-84). */
int st_detector_forward(StDetector* det, const float* img_dev, const float* disp_dev,
                        void* workspace_dev, size_t workspace_bytes, st_stream_t stream,
                        float* head_out_dev);
/* Number of conv MACs one forward performs (for roofline reporting). */
double st_detector_macs(const StDetector* det);
/* Stereo configuration (with_right_branch=1), two phases around the cost-volume module:
 * phase 0 = stem+stage1 features of left AND right (shared RGB-branch weights, one stacked
 * batch of 2N; tap "stage1_rgb" = [2N][H/4][W/4][C]); phase 1 = disparity branch + fused trunk +
 * neck + head.  With with_right_branch=0 the two phases together equal st_detector_forward. */
int st_detector_forward_phase(StDetector* det, int phase, const float* img_dev,
                              const float* disp_dev, const float* right_dev, void* workspace_dev,
                              size_t workspace_bytes, st_stream_t stream, float* head_out_dev);
/* Phase 0 from RAW frames (see st_stem_focus_conv_u8): left / right = HOST arrays of `batch` device pointers to uint8
 * [3][h][w] frames; the detector's height x width is the padded size.  Requires the fused stem (always the case for the
 * shipped widths).  Results are bit-identical to st_pack_raw_frames + st_detector_forward_phase(det, 0, ...). */
int st_detector_forward_phase0_raw(StDetector* det, const unsigned char* const* left_frames_host,
                                   const unsigned char* const* right_frames_host, int h, int w, float pad_value,
                                   void* workspace_dev, size_t workspace_bytes, st_stream_t stream);
/* st_detector_forward (the disparity-INPUT configuration, the reference's shipped one) with the image as RAW frames:
 * img_frames_host = HOST array of `batch` device pointers to uint8 [3][h][w] frames, converted + padded inside the RGB
 * stem (see st_stem_focus_conv_u8); disp_dev = the fp32 (N,3,H,W) disparity input as before.  Bit-identical to
 * st_pack_raw_frames + st_detector_forward. */
int st_detector_forward_raw(StDetector* det, const unsigned char* const* img_frames_host, int h, int w,
                            float pad_value, const float* disp_dev, void* workspace_dev, size_t workspace_bytes,
                            st_stream_t stream, float* head_out_dev);
/* Per-op timing for bench.py / profiling.  When enabled every op (focus pack, conv, spp) of the
 * following forwards is bracketed by hipEvents on the caller's stream; st_detector_op_times
 * synchronises on them and returns, per op: elapsed ms, kind (0 focus, 1 conv, 2 spp), the kernel
 * instance that ran it (0..21 implicit-GEMM tiles, 40 fused stem, 41 streaming 1x1, 42 direct 3x3,
 * 43 / 44 Winograd, 45 fused front, 46 resident 1x1, 47 head prediction, 48 / 49 grouped Winograd launch
 * and its riders, 50..55 split-operand instances; names: st_conv_variant_name; -1 otherwise), conv MACs, phase. */
int st_detector_set_timing(StDetector* det, int enable);
int st_detector_num_ops(const StDetector* det);
int st_detector_op_times(StDetector* det, int cap, float* ms, int* kind, int* variant, double* macs,
                         int* phase);
int st_detector_op_desc(const StDetector* det, int i, char* buf, int cap);
/* Measure every valid conv tile variant on every conv op's real shape and keep the fastest
 * (host-synchronous; call once after st_detector_finalize, never inside a timed region). */
/* Split-operand instances (tile variants 50-55 of st_conv2d_nhwc_variant): the same implicit GEMM with every fp32 operand
 * split into three bf16 terms (error-free: 3 x 8 = 24 mantissa bits), six exact term products on
 * v_mfma_f32_32x32x16_bf16, fp32 accumulate.  PARKED (round 5): the plan built from them did not pass the frozen parity
 * gate (profiles/r05_gpu_tests_split_plan.log) and bought +0.7 % in flight, so they exist in the TOOLS build only
 * (make ABLATION=1).  st_split_instances_available() = 1 there, 0 in the product library, where
 * st_detector_set_split(det, allow != 0) and variants 50-55 return an error; st_detector_set_split(det, 0) is a no-op. */
int st_split_instances_available(void);
int st_detector_set_split(StDetector* det, int allow);
int st_detector_autotune(StDetector* det, void* workspace_dev, size_t workspace_bytes,
                         float* head_out_dev, st_stream_t stream, int reps);
const char* st_conv_variant_name(int id);
const char* st_conv_variant_signature(int id); /* template args of the variant's kernel (profiler row matching) */
/* Per-op tile choice (one int per op of st_detector_num_ops, -1 = heuristic): read it after an
 * autotune, restore it in another process to skip the measurement. */
int st_detector_get_tuning(const StDetector* det, int* variants, int cap);
int st_detector_set_tuning(StDetector* det, const int* variants, int n);
/* Internal NHWC activations inside the workspace (valid after forward); name in
 * {"stage1_rgb","stage1_fused","stage2","stage3","stage4","p3","p4","p5"}.  Pixel p, channel c
 * is ptr[p*ld + c]. */
int st_detector_tap(const StDetector* det, const char* name, const void* workspace_dev,
                    const float** ptr_dev, int* N, int* C, int* H, int* W, int* ld);

/* ------------------------------------------------------------------------
 * 4. Decode + score filter + sort + NMS.
 *    Replaces mmyolo YOLOXHead.predict_by_feat -> YOLOXBBoxCoder.decode ->
 *    mmdet filter_scores_and_topk -> mmcv.ops.batched_nms
 *    (un-vendored; call site reference yolo_detector_disparity_v1.py:121-122,
 *    thresholds configs/stereo_tracking/ocsort/yolox_s_mmyolo_mot_airdrone_disp.py:42).
 * ---------------------------------------------------------------------- */
typedef struct StDecodeDesc {
  int struct_size;
  int batch;
  int num_levels;
  int level_h[4], level_w[4], level_stride[4];
  size_t level_offset[4]; /* float offset of level l inside head_out */
  float score_thr;        /* keep score > thr  */
  float iou_thr;          /* suppress IoU > thr */
  int max_det;            /* capacity of the out_* arrays per image */
  float scale_x, scale_y; /* scale_factor (w,h); boxes /= scale before NMS */
  float pad_left, pad_top;/* pad_param subtracted before scaling (0 if none) */
  float ori_w, ori_h;     /* clamp range after NMS */
  int nms_mask_rows;      /* candidates (in score order) whose pairwise IoU bits are precomputed chip-wide;
                           * 0 = default 4096.  Sizes the workspace (rows^2 / 8 bytes per image); later
                           * candidates are resolved on the fly by one wave - results do not depend on it */
  int num_classes;        /* 0 / 1: one class (the shipped config).  2..1024: head rows carry num_classes class logits
                           * (then x, y, w, h, obj; st_head_row_floats(num_classes) floats per row); multi_label
                           * decode - every (prior, class) pair with score > thr is a candidate, in
                           * filter_scores_and_topk's order - and class-aware NMS by mmcv batched_nms's offset trick
                           * (boxes + label * (max coordinate + 1)); out_labels = class, out_prior_idx = prior */
  int single_label;       /* with several classes: test_cfg.multi_label = False - ONE candidate per prior, the class of
                           * its largest score sigmoid(cls_c) * sigmoid(obj) (first maximum on ties; mmyolo
                           * predict_by_feat: scores.max(1)), thresholded afterwards; NMS stays class-aware */
} StDecodeDesc;

/* Floats per prior in head_out: 8 for 1..3 classes (cls.., x, y, w, h, obj, padding), num_classes + 5 rounded up to a
 * multiple of 4 beyond. */
int st_head_row_floats(int num_classes);

size_t st_decode_nms_workspace_bytes(const StDecodeDesc* d);
/* outputs per image n: out_boxes[n][max_det][4] (xyxy), out_scores[n][max_det],
 * out_labels[n][max_det] (int64), out_prior_idx[n][max_det] (flat prior index,
 * the bit-exact identity of a kept box), out_count[n] = number kept (may exceed
 * max_det: then only the first max_det are stored - the reference applies NO cap under
 * yolox_style=True, so callers must treat out_count[n] > max_det as an overflow and re-run with a
 * larger buffer or raise; rows past min(count, max_det) are written as zero, prior index -1). */
int st_decode_nms(const StDecodeDesc* d, const float* head_out_dev, void* workspace_dev,
                  size_t workspace_bytes, st_stream_t stream, float* out_boxes_dev,
                  float* out_scores_dev, int64_t* out_labels_dev, int32_t* out_prior_idx_dev,
                  int32_t* out_count_dev);

/* ------------------------------------------------------------------------
 * 5. Stereo cost volume -> soft-argmin disparity (NEW module named by
 *    north_star; no reference function exists — consumer contract is
 *    reference loading_disparity.py:129-134 and ocsort_disparity.py:115,132-134).
 *    featL/featR: NHWC [N][Hf][Wf][C].  cost[d] = (1/C) sum_c L[x,c]*R[x-d,c]
 *    (0 where x-d<0); disp_lowres = sum_d d*softmax_d(temperature*cost[d]).
 *    out_cost_dev (optional) receives the materialised volume [N][Hf][Wf][D].
 * ---------------------------------------------------------------------- */
int st_costvolume_softargmin(const float* featL_dev, const float* featR_dev, int N, int Hf,
                             int Wf, int C, int feat_ld, int D, float temperature,
                             float* out_cost_dev, float* out_disp_dev, st_stream_t stream);
/* One 3-D aggregation layer on the materialised volume [N][Hf][Wf][D] (north_star: "its 3D/2D aggregation"; no
 * reference function - the specification is oracle/st_oracle.c::oracle_agg3d, bit-exact): a single-channel 3x3x3
 * convolution over (d, y, x), zero padded, out = act(bias + sum w[kD][kH][kW] * vol[d+kD-1, y+kH-1, x+kW-1]),
 * act = SiLU (1) or none (0).  weight27_host: the 27 taps in (kD, kH, kW) order, HOST memory (they travel in the
 * kernel arguments).  vol_in != vol_out, 16-byte aligned, D a multiple of 4 (<= 192). */
int st_volume_agg3d(const float* vol_in_dev, float* vol_out_dev, int N, int Hf, int Wf, int D,
                    const float* weight27_host, float bias, int act, st_stream_t stream);
/* Cost volume and the FIRST 3-D aggregation layer in one pass (the volume between them never reaches memory):
 * vol_out = agg3d(costvolume(featL, featR)), cell for cell the arithmetic of st_costvolume_softargmin's volume followed by
 * st_volume_agg3d (specification: oracle_costvolume then oracle_agg3d, bit-exact).  featL/featR: NHWC [N][H][W][ld],
 * channels [0, C) used.  Built for the full-resolution mode of the stereo module (north_star's D = 192 sizing; no
 * reference function, consumer contract as for st_costvolume_softargmin).  C in {4, 8, 16}, D a multiple of 4 (<= 192):
 * st_costvolume_agg3d_supported(C, D) tells (1/0); other shapes take the two calls above. */
int st_costvolume_agg3d_supported(int C, int D);
int st_costvolume_agg3d(const float* featL_dev, const float* featR_dev, int N, int H, int W, int C, int feat_ld, int D,
                        const float* weight27_host, float bias, int act, float* vol_out_dev, st_stream_t stream);
/* Round 6: the same pass with the SOFT-ARGMIN of the aggregated volume taken inside the kernel - for a stereo module whose
 * only 3-D layer this is (StereoCostVolume(full_res=True, agg3d_layers=1)): the D-level volume (4 D bytes per pixel, 5.8 GB
 * per 8 pairs at D = 192 x 736 x 1280) is neither written nor read back.  disp_out_dev [N][H][W] px (float), as
 * st_softargmin writes it; bit-equal to st_costvolume_agg3d followed by st_softargmin (specification
 * oracle_costvolume -> oracle_agg3d -> oracle_softargmin).  Needs st_costvolume_agg3d_supported(C, D) and D in {48, 96, 192}
 * (ST_ERR_INVALID otherwise: the caller takes st_costvolume_agg3d + st_softargmin, same results). */
int st_costvolume_agg3d_softargmin(const float* featL_dev, const float* featR_dev, int N, int H, int W, int C, int feat_ld,
                                   int D, const float* weight27_host, float bias, int act, float temperature,
                                   float* disp_out_dev, st_stream_t stream);
/* soft-argmin only, on an (aggregated) volume [N][Hf][Wf][D] */
int st_softargmin(const float* cost_dev, int N, int Hf, int Wf, int D, float temperature,
                  float* out_disp_dev, st_stream_t stream);
/* bilinear (align_corners=False) x`scale` upsample of the low-res disparity,
 * multiplied by `scale`, cropped to (valid_h, valid_w), zero elsewhere, written
 * replicated into the 3 channels of disp_postp NCHW [N][3][H][W]. */
int st_disp_upsample_pack(const float* disp_lr_dev, int N, int Hf, int Wf, int scale, int H,
                          int W, int valid_h, int valid_w, float* disp_postp_dev,
                          st_stream_t stream);

/* Bilinear x`scale` upsampling (align_corners=False) of an NHWC feature map [N][Hf][Wf][C] (pixel stride feat_ld floats)
 * to [N][Hf*scale][Wf*scale][C] dense: the feature side of the stereo module's FULL-RESOLUTION mode
 * (StereoCostVolume(full_res=True): the D = max_disp level volume of north_star's sizing, D x H x W, built at image
 * resolution from reduced + upsampled stage-1 features).  NEW, no reference function; specification
 * oracle/st_oracle.c::oracle_feat_upsample (bit-exact).  Consumer contract of the module's output as for
 * st_disp_upsample_pack: mmtrack/datasets/transforms/loading_disparity.py:85-86,129-134. */
int st_feat_upsample(const float* feat_dev, int N, int Hf, int Wf, int C, int feat_ld, int scale,
                     float* out_dev, st_stream_t stream);

/* ------------------------------------------------------------------------
 * 6. Per-box depth (disp2depth + extract_depth + scale), reference
 *    mmtrack/models/mot/ocsort_disparity.py:113-175 and
 *    mmtrack/models/trackers/utils.py:58-73.
 *    disp: channel 0 of disp_postp, [N][H][W] with row pitch W and image pitch
 *    img_pitch floats.  boxes [N][max_det][4], counts[N].
 *    outputs: depth[N][max_det] (-1 = no valid depth), scale[N][max_det],
 *    scaled_boxes[N][max_det][4].
 * ---------------------------------------------------------------------- */
size_t st_box_depth_workspace_bytes(int N, int max_det, int H, int W);
int st_box_depth(const float* disp_dev, size_t img_pitch, int N, int H, int W,
                 const float* boxes_dev, const int32_t* counts_dev, int max_det, float baseline,
                 float focal, void* workspace_dev, size_t workspace_bytes, st_stream_t stream,
                 float* out_depth_dev, float* out_scale_dev, float* out_scaled_boxes_dev);

/* Frame records: the fixed-size, self-describing unit of the detection all-gather and of the ONE
 * device->host copy per chunk of the MOT shell: out (N, max_det + 1, cols) fp32, row 0 = [true count (may exceed
 * max_det = overflow), max_det, valid-frame flag, 0...], rows 1.. = x1,y1,x2,y2,score,label,depth,scale.
 * mode 0: unscaled boxes (what reference mmtrack/models/mot/ocsort_disparity.py:107-108 returns as pred_det_instances),
 * mode 1: the depth-scaled boxes the tracker consumes (:82-86), cols = 8; mode 2: unscaled box first, scaled box and
 * kept prior index appended, cols = 13.  Frames >= n_real are batch padding (header all zero).  One launch. */
int st_pack_records(const float* boxes_dev, const float* scores_dev, const int64_t* labels_dev, const float* depth_dev,
                    const float* scales_dev, const float* scaled_boxes_dev, const int32_t* prior_idx_dev,
                    const int32_t* counts_dev, int N, int max_det, int mode, int n_real, float* out_records_dev,
                    st_stream_t stream);
/* rows k >= min(counts[n], max_det) of the three outputs are written as 0 (never stale). */

/* ------------------------------------------------------------------------
 * 7. Linear assignment of the CPU association step (HOST function, no GPU):
 *    replaces `lap.lapjv(dists, extend_cost=True, cost_limit=1 - match_iou_thr)`,
 *    reference mmtrack/models/trackers/ocsort_tracker_disparity.py:260-261, :312-313.
 *    cost: row-major float64 [n_rows][n_cols] (tracks x detections).
 *    x_out[n_rows] = column matched to row i or -1; y_out[n_cols] = row matched to
 *    column j or -1.  Dense Jonker-Volgenant on lap's (n_rows+n_cols)^2 extension, so
 *    that non-unique optima resolve as they do there.  NaN costs are unmatchable.
 * ---------------------------------------------------------------------- */
int st_lapjv_extended(const double* cost, int n_rows, int n_cols, double cost_limit,
                      int32_t* x_out, int32_t* y_out);

/* ------------------------------------------------------------------------
 * 8. The whole association step of one frame as a HOST routine (no GPU): the
 *    consumer of the detection records.  Reference
 *    OCSORTTracker_Disparity.track, mmtrack/models/trackers/ocsort_tracker_disparity.py:345-618
 *    (+ kalman_tracker_base.py:49-88, base_tracker.py:54-141, motion/kalman_filter.py:60-189);
 *    constructor kwargs = configs/stereo_tracking/ocsort/yolox_s_mmyolo_mot_airdrone_disp.py:49-58.
 *    dets[n][8] = x1,y1,x2,y2 (depth-SCALED box), score, label, depth, scale - one row of a frame
 *    record; out_rows / out_ids = pred_track_instances in the reference's output order
 *    (matched detections stage by stage, then the newly started tracks).
 * ---------------------------------------------------------------------- */
typedef struct StTracker StTracker;
typedef struct StTrackerConfig {
  int struct_size;
  float obj_score_thr, init_track_thr;
  int weight_iou_with_det_scores;
  float match_iou_thr;
  int num_tentatives;
  float vel_consist_weight;
  int vel_delta_t;
  int num_frames_retain;
} StTrackerConfig;
int st_tracker_create(const StTrackerConfig* cfg, StTracker** out);
int st_tracker_destroy(StTracker* t);
int st_tracker_reset(StTracker* t);
int st_tracker_track(StTracker* t, int frame_id, const float* dets, int n, float* out_rows,
                     int64_t* out_ids, int cap, int* out_n);
/* The same over a CHUNK of frame records (st_pack_records mode 2: (F, rows_per_frame, cols >= 12) fp32 in host memory,
 * e.g. the page-locked buffer the chunk's ONE device->host copy landed in): for every valid frame f the detections
 * (depth-scaled box, score, label, depth, scale) feed st_tracker_track with frame_ids[f]; out_rows (F, cap, 8) /
 * out_ids (F, cap) receive pred_track_instances with the box UNSCALED again (scale_bbox(b, 1 / scale), reference
 * mmtrack/models/mot/ocsort_disparity.py:88-97); out_counts[f] = rows written, -1 for batch-padding frames.  A frame
 * whose record says count > capacity returns ST_ERR_WORKSPACE (the DetectionOverflow of the Python side). */
int st_tracker_track_records(StTracker* t, const int* frame_ids, const float* records, int F, int rows_per_frame,
                             int cols, float* out_rows, int64_t* out_ids, int cap, int* out_counts);
/* state inspection (tests, checkpointing): live tracks in creation order */
int st_tracker_num_tracks(const StTracker* t);
long long st_tracker_next_id(const StTracker* t);
int st_tracker_get_track(const StTracker* t, int index, int64_t* id, double* mean8, double* cov64,
                         int* tentative, int* tracked, int* last_frame);


/* ------------------------------------------------------------------------
 * 9. Batched GPU association: the SAME association step (section 8) for `batch` independent
 *    sequences advanced in lockstep on the device, one wave per sequence and frame; state stays in device memory
 *    between steps.  Results (ids, rows, order) equal st_tracker_track's on the same detections.  For many short
 *    sequences per step (multi-camera serving); for one video the host routine is the faster one.
 *    dets (batch, max_dets, 8) = rows of frame records (depth-scaled box, score, label, depth, scale), counts[b] = rows
 *    of sequence b in this step (-1: sequence b has no frame in this step), frame_ids[b] (0 resets sequence b).
 *    state / scratch: caller-owned device buffers of st_batched_tracker_{state,scratch}_bytes (state zero-filled
 *    before the first step).  out_rows (batch, max_dets, 8), out_ids (batch, max_dets), out_counts (batch),
 *    status (batch): 0 ok, 1 = more than max_tracks live tracks, 2 = counts[b] > max_dets (that sequence's
 *    output count is 0; the caller checks status).  A non-zero status is STICKY: the overflow is detected after the
 *    association has mutated the sequence's tracks, so the sequence is invalid from then on and every later step
 *    reports the same status with an output count of 0 until a step with frame_id 0 resets it.
 *    Enqueued on `stream`, no host sync.
 * ---------------------------------------------------------------------- */
typedef struct StBatchedTracker StBatchedTracker;
int st_batched_tracker_create(const StTrackerConfig* cfg, int batch, int max_tracks, int max_dets,
                              StBatchedTracker** out);
int st_batched_tracker_destroy(StBatchedTracker* t);
size_t st_batched_tracker_state_bytes(const StBatchedTracker* t);
size_t st_batched_tracker_scratch_bytes(const StBatchedTracker* t);
int st_batched_tracker_step(StBatchedTracker* t, const int32_t* frame_ids_dev, const float* dets_dev,
                            const int32_t* counts_dev, void* state_dev, void* scratch_dev, float* out_rows_dev,
                            int64_t* out_ids_dev, int32_t* out_counts_dev, int32_t* status_dev, st_stream_t stream);

/* ----------------------------------------------------------------------
 * Dataset reader helper (host, no GPU): reverse the PNG scanline filters (RFC 2083 6: None/Sub/Up/Average/Paeth).
 * Replaces the OpenCV PNG decode behind mmcv.imfrombytes(..., flag='unchanged') that the reference's loaders call
 * (mmtrack/datasets/transforms/loading_disparity.py:74-75 uint16 disparity, :213-215 uint16 depth; mmcv's
 * LoadImageFromFile for the uint8 left / right images).  Container parsing and inflate stay in the host language
 * (Python zlib, stereotracking_amd/datasets.py).
 *   filtered: height rows of (1 filter-type byte + stride data bytes), as inflate yields them;
 *   bpp: bytes per complete pixel (1 gray8, 2 gray16, 3 rgb8, 4 rgba8, 6 rgb16, 8 rgba16); out: height x stride.
 * ---------------------------------------------------------------------- */
int st_png_unfilter(const uint8_t* filtered, int height, int stride, int bpp, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif /* STEREOTRACK_H_ */
