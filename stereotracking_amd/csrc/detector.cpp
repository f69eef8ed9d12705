// Graph builder + executor for the two-branch YOLOX-s detector
// (CSPDarknet RGB/disparity branches -> PAFPN -> decoupled head).
//
// Restates, as a static launch plan of fused HIP convolutions, what the reference builds
// from Python modules:
//   backbone  mmtrack/models/backbones/csp_darknet_disparity_v1.py:71-206
//             (+ base class det_backbone/base_backbone_disparity_mmyolo.py:76-119)
//   detector  mmtrack/models/detectors/yolo_detector_disparity_v1.py:77-142
//   neck/head mmyolo 0.2.0 YOLOXPAFPN / YOLOXHeadModule (un-vendored),
//             configured at configs/_base_/yolox_s_8x8_mmyolo.py:30-51.
// Parameter names are the reference state_dict keys so checkpoints load unchanged.
//
// Fusions (each line = ONE conv launch):
//   * ConvModule = conv + folded BN + SiLU
//   * CSPLayer main_conv + short_conv (same input)      -> one conv, split store
//   * bottleneck residual add                            -> epilogue
//   * channel concat (CSP, PAFPN)                        -> channel-offset stores into the cat buffer
//   * PAFPN nearest x2 upsample                          -> replicated epilogue store
//   * branch average (rgb + disp)/2                      -> epilogue of disp_stage1's final conv
//   * head cls-tower conv0 + reg-tower conv0             -> one conv, split store
//   * head conv_cls + conv_reg + conv_obj of all three levels -> one reduction launch (head_pred.hip)
//   * stage1.0 (3x3/s2) + stage1.1 main|short + blocks.0.conv1 of both branches -> one launch (front_fused.hip)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

#include "st_common.h"

namespace st {

int conv2d_launch(const StConvDesc& d, hipStream_t stream, int force_variant, int* picked_variant);
bool pw_conv_applicable(const StConvDesc& d);
bool pw_chain_applicable(const StConvDesc& d, const StConvDesc& c);
bool dc_conv_applicable(const StConvDesc& d);
bool wino_conv_applicable(const StConvDesc& d);
bool wino_shape_ok(int Cin, int Cout);
bool wino_group_applicable(const StConvDesc* d, int n);
int wino_group_launch(const StConvDesc* d, int n, hipStream_t stream);
size_t wino_packed_floats(int Cout, int Cin);
int wino_pack_weights(const float* packed, int Cout, int Cin, float* out);
int pw_conv_launch(const StConvDesc& d, hipStream_t stream, const StConvDesc* chain);
bool pwr_conv_applicable(const StConvDesc& d);
bool pwr_chain_applicable(const StConvDesc& d, const StConvDesc& c);
int pwr_conv_launch(const StConvDesc& d, hipStream_t stream, const StConvDesc* chain);
bool front_fused_applicable(const StConvDesc& a, const StConvDesc& ms, const StConvDesc& c1);
int front_fused_launch(const StConvDesc& a, const StConvDesc& ms, const StConvDesc& c1, const float* frag_ms_dev,
                       const float* frag_c1_dev, hipStream_t stream);
size_t front_frag_floats(int Cout, int Cin);
int front_pack_frags(const float* packed, int Cout, int Cin, float* out);
bool csp_tail_applicable(const StConvDesc& c2, const StConvDesc& f);
int csp_tail_launch(const StConvDesc& c2, const StConvDesc& f, const float* frag_fin_dev, hipStream_t stream);
size_t csp_tail_frag_floats();
int csp_tail_pack_frags(const float* packed, float* out);
int conv_variant_count();
bool conv_variant_valid(int id, int cout);
const char* conv_variant_name(int id);
const char* conv_variant_signature(int id);
int focus_pack_launch(const float* img, int N, int C, int H, int W, float* out, hipStream_t stream);
int stem_focus_conv_launch(const float* in, int N, int H, int W, int used_planes, const float* wgt,
                           const float* bias, int Cout, float* out, int out_ld, int out_off, int act,
                           hipStream_t stream, const StemRawInput* raw);
int spp_pool_launch(const float* x, int x_ld, int x_off, int N, int H, int W, int C, float* out,
                    int out_ld, int out_off, hipStream_t stream);
struct HeadPredLevel {
  const float *cls, *reg, *wc, *wr, *bc, *br;
  float* out;
  int cls_ld, cls_off, reg_ld, reg_off, M, blk0;
};
struct HeadPredArgs {
  HeadPredLevel lv[3];
  int Kpad, nc, nblocks;
};
bool head_pred_applicable(int feat, int nc);
int head_pred_launch(HeadPredArgs a, int feat, hipStream_t stream);

namespace {

int make_divisible(double x, double widen) { return (int)std::ceil(x * widen / 8.0) * 8; }
int make_round(int n, double deepen) {
  if (n <= 1) return n;
  // python round() = banker's rounding; x.5 never occurs for the shipped factors but keep it exact
  const double v = n * deepen;
  double r = std::nearbyint(v);
  return std::max((int)r, 1);
}

struct Param {
  std::string name;
  std::vector<int64_t> shape;
  std::vector<float> data;
  bool set = false;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

struct ConvSrc {
  std::string conv_prefix;  // "<p>.conv" for ConvModule, "<p>" for a bare Conv2d
  std::string bn_prefix;    // "" = no BN
  bool has_bias;
  int cout;
};

struct PackedConv {
  std::vector<ConvSrc> srcs;
  int cin = 0, k = 1, cout = 0;
  size_t wgt_off = 0, bias_off = 0;  // float offsets inside the packed weight arena
  size_t wino_off = 0;               // Winograd-form copy of the weights (3x3 / stride-1 users only), 0 = none
  bool wino = false;
  size_t frag_off = 0;               // MFMA-fragment-ordered copy of a 1x1 weight matrix (fused front kernel), 0 = none
  bool frag = false;
  size_t tail_off = 0;               // fragment-ordered copy of a 64 x 64 final_conv matrix (fused CSP tail, wino_csp_tail.hip)
  bool tail = false;
  bool stem = false;                 // fused Focus+stem layout (st_stem_pack_weights), cin = 12, k = 3
  int stem_planes = 3;               // image planes the fused stem reads (1: identical planes, summed weights)
};

constexpr int BUF_HEAD = -2;   // caller's head_out buffer
constexpr int BUF_NONE = -1;

struct TRef {  // a channel slice of an NHWC buffer
  int buf = BUF_NONE;
  int N = 0, H = 0, W = 0, C = 0;
  int ld = 0, off = 0;
  size_t base = 0;  // extra float offset inside the buffer (head levels)
  bool adv = false; // full-batch buffer viewed through a sub-batch window: pointer advances per sub-batch
  bool valid() const { return buf != BUF_NONE; }
  TRef slice(int o, int c) const {
    TRef t = *this;
    t.off = off + o;
    t.C = c;
    return t;
  }
};

struct Op {
  enum Type { FOCUS, CONV, SPP, STEM, PRED } type;
  // FOCUS / STEM: src input index (0 = img/left, 1 = disp, 2 = right), dst tensor, dst batch offset
  // (STEM = fused Focus + stem ConvModule, stem_focus_conv.hip; uses pc and out1)
  int focus_input = 0;
  int focus_batch_off = 0;
  // CONV
  int pc = -1;
  TRef in, out1, out2, up, res;
  int split = 0, stride = 1, pad = 0, act = 1;
  float post_scale = 1.f;
  // SPP uses in (x) and out1 (cat buffer, x lives in its first C channels)
  int phase = 0;
  double macs = 0.0;   // conv MACs of this op
  int variant = -1;    // conv tile variant picked at the last launch
  int tuned = -1;      // measured best variant (st_detector_autotune), -1 = heuristic
  int group = 0;       // sub-batch group (0 = whole batch in one launch)
  bool chain_next = false;  // the NEXT op is a 1x1 conv on this op's out1 (CSP main_conv -> bottleneck conv1):
                            // when both run on the streaming kernel (variant 41), or both on the LDS-resident
                            // kernel (variant 46), they are launched as one
  // PRED (head_pred.hip): per level the two tower outputs, the packed conv_cls / conv_reg|obj and the head rows
  TRef pred_cls[3], pred_reg[3], pred_out[3];
  int pred_pcc[3] = {-1, -1, -1}, pred_pcr[3] = {-1, -1, -1};
  int wgroup = 0;           // > 0: consecutive ops with the same wgroup are INDEPENDENT 3x3 convs (head towers of the
                            // three levels) and run as ONE grouped Winograd launch when all of them are tuned to it
  bool tail_next = false;   // this bottleneck conv2 (3x3, 32 -> 32, + identity) and the NEXT op (the CSP final_conv, 64 -> 64)
                            // are one persistent launch of wino_csp_tail.hip (variant 56) whenever both descriptors qualify
  bool front_next2 = false; // this 3x3/s2 conv and the NEXT TWO ops (CSP main|short, blocks.0.conv1) are one launch
                            // of front_fused.hip (variant 45) whenever the three descriptors qualify
};

}  // namespace

}  // namespace st

using namespace st;

struct StDetector {
  StDetectorConfig cfg{};
  std::vector<Param> params;
  std::map<std::string, int> pindex;
  std::vector<PackedConv> convs;
  std::vector<Op> ops;
  std::vector<size_t> buf_off;  // float offsets of workspace buffers
  size_t ws_floats = 0;
  size_t wgt_floats = 0;
  float* wgt_dev = nullptr;
  bool finalized = false;
  double macs = 0.0;
  // head
  int n_levels = 3;
  int lvl_h[3]{}, lvl_w[3]{}, lvl_stride[3]{};
  size_t lvl_off[3]{};
  size_t head_floats = 0;
  StemRawInput raw_inputs[3] = {{nullptr, 0, 0, 0.f}, {nullptr, 0, 0, 0.f}, {nullptr, 0, 0, 0.f}};   // set for the duration of st_detector_forward_phase_raw
  std::map<std::string, TRef> taps;
  int cur_phase = 0;
  // optional per-op timing (bench/profiling only): events on the caller's stream around every op
  bool timing = false;
  std::vector<hipEvent_t> events;  // 2 per op
  int force_variant = -1;          // autotune only
  bool no_wino = false;            // keep the autotuner off the Winograd instance (exact-MFMA-order A/B runs)
  bool allow_chain = true;         // fuse CSP main_conv -> bottleneck conv1 when both run on the streaming kernel
  bool allow_front = true;         // fuse stage1.0 -> main|short -> conv1 (front_fused.hip)
  bool allow_tail = true;          // fuse the last bottleneck conv2 -> final_conv of the stage-1 CSP layers (wino_csp_tail.hip)
  bool allow_wgroup = true;        // head tower convs of the three levels as grouped Winograd launches
  int allow_split = 0;             // bit i: autotune may pick the split-operand (bf16x3) instance 50 + i (st_detector_set_split)
#ifdef ST_ABLATION
  std::vector<char> skip;          // tools-only: ops whose launches are dropped (st_detector_set_skip)
#endif
  // Sub-batch groups: the high-resolution front of the network is run SB images at a time through
  // SB-sized intermediate buffers that are REUSED by every sub-batch, so the intermediates of one
  // sub-batch (~100 MB per image at 736x1280) stay in the 256 MiB Infinity Cache instead of
  // streaming N x that through HBM.  groups[g] = {SB, count}; group 0 = none.
  struct Group { int sb, count; };
  std::vector<Group> groups{{0, 1}};
  int cur_group = 0;
  int max_group_count = 1;
  int begin_group(int sb, int total) {
    groups.push_back({sb, total / sb});
    max_group_count = std::max(max_group_count, total / sb);
    cur_group = (int)groups.size() - 1;
    return cur_group;
  }
  void end_group() { cur_group = 0; }
  static TRef window(TRef full, int sb) {  // sub-batch window of a full-batch tensor
    full.N = sb;
    full.adv = true;
    return full;
  }

  // ---- building blocks -------------------------------------------------------------------
  int add_param(const std::string& name, std::vector<int64_t> shape) {
    Param p;
    p.name = name;
    p.shape = std::move(shape);
    params.push_back(std::move(p));
    pindex[name] = (int)params.size() - 1;
    return (int)params.size() - 1;
  }
  void declare_convmodule(const std::string& p, int cin, int cout, int k) {
    if (pindex.count(p + ".conv.weight")) return;  // shared weights (right branch)
    add_param(p + ".conv.weight", {cout, cin, k, k});
    add_param(p + ".bn.weight", {cout});
    add_param(p + ".bn.bias", {cout});
    add_param(p + ".bn.running_mean", {cout});
    add_param(p + ".bn.running_var", {cout});
  }
  void declare_conv2d(const std::string& p, int cin, int cout, int k) {
    add_param(p + ".weight", {cout, cin, k, k});
    add_param(p + ".bias", {cout});
  }
  int new_buf(int N, int H, int W, int ld) {
    buf_off.push_back(ws_floats);
    ws_floats += (size_t)N * H * W * ld;
    ws_floats = (ws_floats + 63) & ~(size_t)63;  // 256 B alignment
    return (int)buf_off.size() - 1;
  }
  TRef new_tensor(int N, int H, int W, int C) {
    TRef t;
    t.buf = new_buf(N, H, W, C);
    t.N = N; t.H = H; t.W = W; t.C = C; t.ld = C; t.off = 0;
    return t;
  }
  // ConvModule(s) sharing one input, fused along Cout
  int packed_convmodules(const std::vector<std::string>& prefixes, int cin, const std::vector<int>& couts, int k) {
    // reuse when the same single prefix was packed before (right branch shares weights)
    PackedConv pc;
    pc.cin = cin; pc.k = k;
    for (size_t i = 0; i < prefixes.size(); ++i) {
      declare_convmodule(prefixes[i], cin, couts[i], k);
      pc.srcs.push_back({prefixes[i] + ".conv", prefixes[i] + ".bn", false, couts[i]});
      pc.cout += couts[i];
    }
    convs.push_back(pc);
    return (int)convs.size() - 1;
  }
  int packed_conv2d(const std::vector<std::string>& prefixes, int cin, const std::vector<int>& couts) {
    PackedConv pc;
    pc.cin = cin; pc.k = 1;
    for (size_t i = 0; i < prefixes.size(); ++i) {
      declare_conv2d(prefixes[i], cin, couts[i], 1);
      pc.srcs.push_back({prefixes[i], "", true, couts[i]});
      pc.cout += couts[i];
    }
    convs.push_back(pc);
    return (int)convs.size() - 1;
  }
  void op_conv(int pc, const TRef& in, int stride, const TRef& out1, int split = -1,
               const TRef& out2 = TRef(), const TRef& up = TRef(), const TRef& res = TRef(),
               float post_scale = 1.f, int act = 1) {
    Op o;
    o.type = Op::CONV;
    o.pc = pc; o.in = in; o.out1 = out1; o.out2 = out2; o.up = up; o.res = res;
    o.split = split < 0 ? convs[pc].cout : split;
    o.stride = stride; o.pad = convs[pc].k / 2; o.act = act; o.post_scale = post_scale;
    o.phase = cur_phase;
    o.group = cur_group;
    const int Ho = (in.H + 2 * o.pad - convs[pc].k) / stride + 1;
    const int Wo = (in.W + 2 * o.pad - convs[pc].k) / stride + 1;
    o.macs = (double)in.N * groups[cur_group].count * Ho * Wo * convs[pc].k * convs[pc].k * convs[pc].cin *
             convs[pc].cout;
    // wide 3x3 / stride-1 layers also get their weights in Winograd form (kernel instance 43, picked by the autotuner)
    if (convs[pc].k == 3 && stride == 1 && wino_shape_ok(convs[pc].cin, convs[pc].cout) && o.split == convs[pc].cout &&
        !up.valid())
      convs[pc].wino = true;
    macs += o.macs;
    ops.push_back(o);
  }
  // Fused Focus + stem ConvModule on input `input_idx`, written to images [batch_off, batch_off + cfg.batch) of out
  void op_stem(int pc, int input_idx, int batch_off, const TRef& out) {
    Op o;
    o.type = Op::STEM;
    o.pc = pc; o.focus_input = input_idx; o.focus_batch_off = batch_off; o.out1 = out;
    o.phase = cur_phase;
    o.macs = (double)cfg.batch * out.H * out.W * 36.0 * convs[pc].stem_planes * convs[pc].cout;
    o.variant = 40;
    macs += o.macs;
    ops.push_back(o);
  }
  // ConvModule helper: allocates the output unless `out` is given
  TRef convmodule(const std::string& p, const TRef& in, int cout, int k, int stride, TRef out = TRef()) {
    const int pc = packed_convmodules({p}, in.C, {cout}, k);
    if (!out.valid()) {
      const int pad = k / 2;
      out = new_tensor(in.N, (in.H + 2 * pad - k) / stride + 1, (in.W + 2 * pad - k) / stride + 1, cout);
    }
    op_conv(pc, in, stride, out);
    return out;
  }
  // mmdet CSPLayer(cin, cout, n, add_identity), expand_ratio 0.5.  `out` = destination slice of
  // final_conv; res/post_scale/up decorate the final conv's epilogue.
  TRef csp_layer(const std::string& p, const TRef& x, int cout, int nblocks, bool identity,
                 TRef out = TRef(), const TRef& res = TRef(), float post_scale = 1.f,
                 const TRef& up = TRef()) {
    const int mid = cout / 2;
    TRef cat = new_tensor(x.N, x.H, x.W, 2 * mid);
    TRef mainb = new_tensor(x.N, x.H, x.W, mid);
    const int pc = packed_convmodules({p + ".main_conv", p + ".short_conv"}, x.C, {mid, mid}, 1);
    op_conv(pc, x, 1, mainb, mid, cat.slice(mid, mid));
    if (nblocks > 0 && (mid == 32 || mid == 64)) ops.back().chain_next = true;   // main_conv -> blocks.0.conv1 (reads mainb)
    // stage-1 shape (32 -> 64 -> 32|32 -> 32): a preceding 3x3/s2 ConvModule can take both 1x1 convs along
    const bool frontable = nblocks > 0 && mid == 32 && x.C == 64 && ops.size() >= 2 && ops[ops.size() - 2].type == Op::CONV &&
                           convs[ops[ops.size() - 2].pc].k == 3 && ops[ops.size() - 2].stride == 2 &&
                           convs[ops[ops.size() - 2].pc].cin == 32 && ops[ops.size() - 2].out1.buf == x.buf &&
                           ops[ops.size() - 2].group == cur_group && ops[ops.size() - 2].phase == cur_phase;
    if (frontable) {
      ops[ops.size() - 2].front_next2 = true;
      convs[pc].frag = true;
    }
    TRef tmp = new_tensor(x.N, x.H, x.W, mid);
    for (int b = 0; b < nblocks; ++b) {
      const std::string bp = p + ".blocks." + std::to_string(b);
      const int pc1 = packed_convmodules({bp + ".conv1"}, mid, {mid}, 1);
      if (b == 0 && frontable) convs[pc1].frag = true;
      op_conv(pc1, mainb, 1, tmp);
      const int pc2 = packed_convmodules({bp + ".conv2"}, mid, {mid}, 3);
      const bool last = b == nblocks - 1;
      op_conv(pc2, tmp, 1, last ? cat.slice(0, mid) : mainb, -1, TRef(), TRef(),
              identity ? mainb : TRef(), 1.f);
    }
    if (!out.valid()) out = new_tensor(x.N, x.H, x.W, cout);
    const int pcf = packed_convmodules({p + ".final_conv"}, 2 * mid, {cout}, 1);
    // stage-1 shape (conv2 32 -> 32 + identity, final 64 -> 64, no upsampled store): the last bottleneck's conv2 and the
    // final conv can run as one persistent launch (its epilogue takes the residual / post_scale of the final conv)
    const bool tailable = nblocks > 0 && mid == 32 && cout == 64 && identity && !up.valid() && ops.back().type == Op::CONV &&
                          convs[ops.back().pc].k == 3 && convs[ops.back().pc].wino;
    const size_t conv2_idx = ops.size() - 1;
    op_conv(pcf, cat, 1, out, -1, TRef(), up, res, post_scale);
    if (tailable && ops[conv2_idx].group == ops.back().group && ops[conv2_idx].phase == ops.back().phase) {
      ops[conv2_idx].tail_next = true;
      convs[pcf].tail = true;
    }
    return out;
  }

  int build();
};

int StDetector::build() {
  const double w = cfg.widen_factor, dpt = cfg.deepen_factor;
  const int N = cfg.batch, H = cfg.height, W = cfg.width;
  const bool stereo = cfg.with_right_branch != 0;
  const int c1 = make_divisible(64, w), c2 = make_divisible(128, w), c3 = make_divisible(256, w),
            c4 = make_divisible(512, w), c5 = make_divisible(1024, w);
  const int n1 = make_round(3, dpt), n2 = make_round(9, dpt), n3 = make_round(9, dpt),
            n4 = make_round(3, dpt), nn = make_round(3, dpt);
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8, H16 = H / 16,
            W16 = W / 16, H32 = H / 32, W32 = W / 32;

  // neck concat buffers first: backbone stage outputs are stored straight into them
  TRef catTD0 = new_tensor(N, H16, W16, 2 * c4);  // [up(r2) | C4]
  TRef catTD1 = new_tensor(N, H8, W8, 2 * c3);    // [up(t2) | C3]
  TRef catBU0 = new_tensor(N, H16, W16, 2 * c3);  // [down(P3') | t2]
  TRef catBU1 = new_tensor(N, H32, W32, 2 * c4);  // [down(P4') | r2]

  // ---- phase 0: RGB branch stem + stage1 (left, and right when stereo: same weights, batch 2N)
  cur_phase = 0;
  const int NB = stereo ? 2 * N : N;
  // Focus + stem ConvModule as ONE kernel reading the planar image (stem_focus_conv.hip) whenever the stem is
  // at most 64 channels wide (widen_factor <= 1); otherwise focus_pack + the generic conv.
  bool fused_stem = c1 <= 64;
#ifdef ST_ABLATION   // tools-only build: ST_NO_FUSED_STEM=1 forces the two-kernel path (A/B measurements)
  if (getenv("ST_NO_FUSED_STEM")) fused_stem = false;
  if (getenv("ST_NO_FUSED_FRONT")) allow_front = false;
  if (getenv("ST_NO_FUSED_TAIL")) allow_tail = false;
  if (getenv("ST_NO_WINO_GROUP")) allow_wgroup = false;
#endif
  TRef packed_rgb, stem_rgb;
  if (fused_stem) {
    stem_rgb = new_tensor(NB, H2, W2, c1);
    const int pcs = packed_convmodules({"backbone.stem.conv"}, 12, {c1}, 3);
    convs[pcs].stem = true;
    op_stem(pcs, 0, 0, stem_rgb);
    if (stereo) op_stem(pcs, 2, N, stem_rgb);
  } else {
    packed_rgb = new_tensor(NB, H2, W2, 12);
    Op f; f.type = Op::FOCUS; f.focus_input = 0; f.out1 = packed_rgb; f.focus_batch_off = 0; f.phase = 0;
    ops.push_back(f);
    if (stereo) {
      Op g = f; g.focus_input = 2; g.focus_batch_off = N;
      ops.push_back(g);
    }
  }
  // images per sub-batch of the high-resolution front.  Measured on MI355X (bench.py, N=8): off 1135,
  // SB=4 1110, SB=2 1066, SB=1 974 pairs/s - the smaller launches cost more than the Infinity-Cache
  // residency buys, so the default is OFF (the tools-only ST_ABLATION build reads ST_SUBBATCH=k).
  int sbatch = 0;
#ifdef ST_ABLATION
  if (const char* e = getenv("ST_SUBBATCH")) sbatch = atoi(e);
#endif
  if (sbatch <= 0 || N % sbatch != 0) sbatch = N;  // one group covering the whole batch
  TRef s1 = new_tensor(NB, H4, W4, c2);  // stage1 features of every (left | right) image: kept for the stereo module
  int sb0 = sbatch;                      // phase 0 runs left | right as two sub-batches of N (shared intermediates)
#ifdef ST_ABLATION
  if (getenv("ST_MERGE_LR")) sb0 = NB;   // tools: one launch over all 2N images (A/B, profiles/r06_merge_lr_ab.txt)
#endif
  begin_group(sb0, NB);
  {
    TRef stem = fused_stem ? window(stem_rgb, sb0)
                           : convmodule("backbone.stem.conv", window(packed_rgb, sb0), c1, 3, 1);
    TRef s1c = convmodule("backbone.stage1.0", stem, c2, 3, 2);
    csp_layer("backbone.stage1.1", s1c, c2, n1, true, window(s1, sb0));
  }
  end_group();
  taps["stage1_rgb"] = s1;

  // ---- phase 1: disparity branch + everything after the fusion
  cur_phase = 1;
  TRef s1_left = s1;
  s1_left.N = N;  // first N images of the stacked batch
  const bool rgb_only = cfg.rgb_only != 0;   // mmtrack.CSPDarknet (csp_darknet.py:8-13): the image branch alone
  TRef packed_disp, stem_disp;
  if (rgb_only) {
    // no disparity branch: nothing to stage
  } else if (fused_stem) {
    stem_disp = new_tensor(N, H2, W2, c1);
    const int pcs = packed_convmodules({"backbone.disp_stem.conv"}, 12, {c1}, 3);
    convs[pcs].stem = true;
    convs[pcs].stem_planes = cfg.disp_planes_identical ? 1 : 3;
    op_stem(pcs, 1, 0, stem_disp);
  } else {
    packed_disp = new_tensor(N, H2, W2, 12);
    Op f; f.type = Op::FOCUS; f.focus_input = 1; f.out1 = packed_disp; f.focus_batch_off = 0; f.phase = 1;
    ops.push_back(f);
  }
  TRef y = rgb_only ? s1_left : new_tensor(N, H4, W4, c2);
  TRef C3 = catTD1.slice(c3, c3);
  begin_group(sbatch, N);
  {
    if (!rgb_only) {
      TRef dstem = fused_stem ? window(stem_disp, sbatch)
                              : convmodule("backbone.disp_stem.conv", window(packed_disp, sbatch), c1, 3, 1);
      TRef d1c = convmodule("backbone.disp_stage1.0", dstem, c2, 3, 2);
      // y = (o_stem + o_disp_stem) / 2   (csp_darknet_disparity_v1.py:184)
      csp_layer("backbone.disp_stage1.1", d1c, c2, n1, true, window(y, sbatch), window(s1_left, sbatch), 0.5f);
    }
    TRef s2c = convmodule("backbone.stage2.0", window(y, sbatch), c3, 3, 2);
    csp_layer("backbone.stage2.1", s2c, c3, n2, true, window(C3, sbatch));
  }
  end_group();
  taps["stage1_fused"] = y;
  taps["stage2"] = C3;
  TRef s3c = convmodule("backbone.stage3.0", C3, c4, 3, 2);
  TRef C4 = csp_layer("backbone.stage3.1", s3c, c4, n3, true, catTD0.slice(c4, c4));
  taps["stage3"] = C4;
  TRef s4c = convmodule("backbone.stage4.0", C4, c5, 3, 2);
  // SPPFBottleneck(c5, c5, kernel_sizes=(5,9,13)): conv1 c5->c5/2, cat 4x, conv2 2*c5->c5
  TRef sppcat = new_tensor(N, H32, W32, 2 * c5);
  convmodule("backbone.stage4.1.conv1", s4c, c5 / 2, 1, 1, sppcat.slice(0, c5 / 2));
  {
    Op o; o.type = Op::SPP; o.in = sppcat.slice(0, c5 / 2); o.out1 = sppcat; o.phase = 1;
    ops.push_back(o);
  }
  TRef spp = convmodule("backbone.stage4.1.conv2", sppcat, c5, 1, 1);
  TRef C5 = csp_layer("backbone.stage4.2", spp, c5, n4, false);
  taps["stage4"] = C5;

  // ---- neck: YOLOXPAFPN(in=[c3,c4,c5], out=c3)
  const int outc = make_divisible(256, w);
  {
    // r2 = reduce_layers.2 : c5 -> c4, stored into catBU1[c4:] and x2-upsampled into catTD0[:c4]
    const int pc = packed_convmodules({"neck.reduce_layers.2"}, c5, {c4}, 1);
    op_conv(pc, C5, 1, catBU1.slice(c4, c4), -1, TRef(), catTD0.slice(0, c4));
  }
  // top_down_layers.0 = Sequential(CSP(2*c4 -> c4), ConvModule(c4 -> c3, 1))
  TRef td0 = csp_layer("neck.top_down_layers.0.0", catTD0, c4, nn, false);
  {
    const int pc = packed_convmodules({"neck.top_down_layers.0.1"}, c4, {c3}, 1);
    op_conv(pc, td0, 1, catBU0.slice(c3, c3), -1, TRef(), catTD1.slice(0, c3));  // t2
  }
  TRef P3 = csp_layer("neck.top_down_layers.1", catTD1, c3, nn, false);
  taps["p3_inner"] = P3;
  convmodule("neck.downsample_layers.0", P3, c3, 3, 2, catBU0.slice(0, c3));
  TRef P4 = csp_layer("neck.bottom_up_layers.0", catBU0, c4, nn, false);
  convmodule("neck.downsample_layers.1", P4, c4, 3, 2, catBU1.slice(0, c4));
  TRef P5 = csp_layer("neck.bottom_up_layers.1", catBU1, c5, nn, false);
  TRef F[3];
  F[0] = convmodule("neck.out_layers.0", P3, outc, 1, 1);
  F[1] = convmodule("neck.out_layers.1", P4, outc, 1, 1);
  F[2] = convmodule("neck.out_layers.2", P5, outc, 1, 1);
  taps["p3"] = F[0]; taps["p4"] = F[1]; taps["p5"] = F[2];

  // ---- head: YOLOXHeadModule(in=outc, feat=outc, stacked_convs=2)
  const int feat = make_divisible(256, w);
  const int nc = cfg.num_classes;
  ST_REQUIRE(nc >= 1 && nc <= 1024, "detector: num_classes must be in [1, 1024]");
  const int hr = head_row_floats(nc);   // floats per prior in the head buffer
  const std::string hp = "bbox_head.head_module.";
  head_floats = 0;
  for (int l = 0; l < 3; ++l) {
    lvl_h[l] = F[l].H; lvl_w[l] = F[l].W; lvl_stride[l] = 8 << l;
    lvl_off[l] = head_floats;
    head_floats += (size_t)N * F[l].H * F[l].W * hr;
  }
  bool fused_pred = head_pred_applicable(feat, nc);
#ifdef ST_ABLATION
  if (getenv("ST_NO_FUSED_PRED")) fused_pred = false;
#endif
  Op pred;
  pred.type = Op::PRED; pred.phase = cur_phase; pred.variant = 47;
  // The towers of the three levels are independent of each other: the ops are emitted DEPTH-major (conv0 of every
  // level, then the second convs of every level) and tagged as groups, so that run_ops can put each depth into one
  // grouped Winograd launch (wino_conv.hip: the small maps' workgroups ride in the big map's grid).
  TRef t0s[3], clsfs[3], regfs[3];
  for (int l = 0; l < 3; ++l) {
    const std::string ls = std::to_string(l);
    t0s[l] = new_tensor(N, F[l].H, F[l].W, 2 * feat);  // [cls_feat0 | reg_feat0]
    const int pc0 = packed_convmodules({hp + "multi_level_cls_convs." + ls + ".0",
                                        hp + "multi_level_reg_convs." + ls + ".0"},
                                       outc, {feat, feat}, 3);
    op_conv(pc0, F[l], 1, t0s[l]);
    ops.back().wgroup = 1;
  }
  for (int l = 0; l < 3; ++l) {
    const std::string ls = std::to_string(l);
    clsfs[l] = convmodule(hp + "multi_level_cls_convs." + ls + ".1", t0s[l].slice(0, feat), feat, 3, 1);
    ops.back().wgroup = 2;
    regfs[l] = convmodule(hp + "multi_level_reg_convs." + ls + ".1", t0s[l].slice(feat, feat), feat, 3, 1);
    ops.back().wgroup = 2;
  }
  for (int l = 0; l < 3; ++l) {
    const std::string ls = std::to_string(l);
    TRef clsf = clsfs[l], regf = regfs[l];
    TRef ho;
    ho.buf = BUF_HEAD; ho.N = N; ho.H = F[l].H; ho.W = F[l].W; ho.ld = hr; ho.base = lvl_off[l];
    const int pcc = packed_conv2d({hp + "multi_level_conv_cls." + ls}, feat, {nc});
    const int pcr = packed_conv2d({hp + "multi_level_conv_reg." + ls, hp + "multi_level_conv_obj." + ls},
                                  feat, {4, 1});
    if (fused_pred) {   // all prediction convs of all levels: one launch after the towers
      pred.pred_cls[l] = clsf; pred.pred_reg[l] = regf; pred.pred_out[l] = ho;
      pred.pred_pcc[l] = pcc; pred.pred_pcr[l] = pcr;
      pred.macs += (double)N * F[l].H * F[l].W * feat * (nc + 5);
    } else {
      op_conv(pcc, clsf, 1, ho.slice(0, nc), -1, TRef(), TRef(), TRef(), 1.f, /*act=*/0);
      op_conv(pcr, regf, 1, ho.slice(nc, 5), -1, TRef(), TRef(), TRef(), 1.f, /*act=*/0);
    }
  }
  if (fused_pred) {
    macs += pred.macs;
    ops.push_back(pred);
  }

  // packed weight arena layout
  wgt_floats = 0;
  for (auto& pc : convs) {
    pc.wgt_off = wgt_floats;
    wgt_floats += pc.stem ? st_stem_packed_floats(pc.cout)
                          : (size_t)round_up(pc.cout, 32) * round_up(pc.k * pc.k * pc.cin, 32);
    pc.bias_off = wgt_floats;
    wgt_floats += round_up(pc.cout, 32);
    wgt_floats = (wgt_floats + 63) & ~(size_t)63;
    if (pc.wino) {
      pc.wino_off = wgt_floats;
      wgt_floats += wino_packed_floats(pc.cout, pc.cin);
      wgt_floats = (wgt_floats + 63) & ~(size_t)63;
    }
    if (pc.frag) {
      pc.frag_off = wgt_floats;
      wgt_floats += front_frag_floats(pc.cout, pc.cin);
      wgt_floats = (wgt_floats + 63) & ~(size_t)63;
    }
    if (pc.tail) {
      pc.tail_off = wgt_floats;
      wgt_floats += csp_tail_frag_floats();
      wgt_floats = (wgt_floats + 63) & ~(size_t)63;
    }
  }
  return ST_OK;
}

extern "C" int st_detector_create(const StDetectorConfig* cfg, StDetector** out) {
  if (!cfg || !out) return set_error(ST_ERR_INVALID, "st_detector_create: null argument");
  ST_REQUIRE(cfg->struct_size == (int)sizeof(StDetectorConfig), "st_detector_create: struct_size mismatch (%d vs %zu)",
             cfg->struct_size, sizeof(StDetectorConfig));
  ST_REQUIRE(cfg->batch > 0 && cfg->height > 0 && cfg->width > 0 && cfg->height % 32 == 0 && cfg->width % 32 == 0,
             "st_detector_create: height/width must be positive multiples of 32 (got %dx%d)", cfg->height, cfg->width);
  ST_REQUIRE(cfg->widen_factor > 0 && cfg->deepen_factor > 0, "st_detector_create: bad widen/deepen factor");
  auto det = std::make_unique<StDetector>();
  det->cfg = *cfg;
  if (det->cfg.bn_eps <= 0) det->cfg.bn_eps = 1e-3;
  ST_CHECK(det->build());
  *out = det.release();
  return ST_OK;
}

extern "C" int st_detector_destroy(StDetector* det) {
  if (!det) return ST_OK;
  if (det->wgt_dev) (void)hipFree(det->wgt_dev);
  for (auto& e : det->events) (void)hipEventDestroy(e);
  delete det;
  return ST_OK;
}

extern "C" int st_detector_num_params(const StDetector* det) { return det ? (int)det->params.size() : 0; }

extern "C" int st_detector_param_info(const StDetector* det, int idx, char* name, int name_cap,
                                      int64_t shape[4], int* ndim) {
  if (!det || idx < 0 || idx >= (int)det->params.size())
    return set_error(ST_ERR_INVALID, "st_detector_param_info: bad index %d", idx);
  const Param& p = det->params[idx];
  if (name && name_cap > 0) {
    std::strncpy(name, p.name.c_str(), (size_t)name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (ndim) *ndim = (int)p.shape.size();
  if (shape)
    for (size_t i = 0; i < 4; ++i) shape[i] = i < p.shape.size() ? p.shape[i] : 1;
  return ST_OK;
}

extern "C" int st_detector_set_param(StDetector* det, const char* name, const float* host, int64_t numel) {
  if (!det || !name || !host) return set_error(ST_ERR_INVALID, "st_detector_set_param: null argument");
  auto it = det->pindex.find(name);
  if (it == det->pindex.end()) return set_error(ST_ERR_NOTFOUND, "st_detector_set_param: unknown parameter '%s'", name);
  Param& p = det->params[it->second];
  ST_REQUIRE(numel == p.numel(), "st_detector_set_param: '%s' expects %lld values, got %lld", name,
             (long long)p.numel(), (long long)numel);
  p.data.assign(host, host + numel);
  p.set = true;
  det->finalized = false;
  return ST_OK;
}

extern "C" int st_detector_finalize(StDetector* det) {
  if (!det) return set_error(ST_ERR_INVALID, "st_detector_finalize: null detector");
  for (const Param& p : det->params)
    if (!p.set) return set_error(ST_ERR_STATE, "st_detector_finalize: parameter '%s' was never set", p.name.c_str());
  std::vector<float> host(det->wgt_floats, 0.f);
  auto get = [&](const std::string& n) -> const float* { return det->params[det->pindex.at(n)].data.data(); };
  for (const PackedConv& pc : det->convs) {
    if (pc.stem) {
      const ConvSrc& s = pc.srcs[0];
      ST_CHECK(st_stem_pack_weights(get(s.conv_prefix + ".weight"), nullptr, get(s.bn_prefix + ".weight"),
                                    get(s.bn_prefix + ".bias"), get(s.bn_prefix + ".running_mean"),
                                    get(s.bn_prefix + ".running_var"), det->cfg.bn_eps, pc.cout, pc.stem_planes,
                                    host.data() + pc.wgt_off, host.data() + pc.bias_off));
      continue;
    }
    const int Kpad = round_up(pc.k * pc.k * pc.cin, 32);
    int row = 0;
    for (const ConvSrc& s : pc.srcs) {
      const size_t nf = st_conv_packed_floats(s.cout, pc.cin, pc.k, pc.k);
      std::vector<float> wtmp(nf), btmp(round_up(s.cout, 32));
      const bool bn = !s.bn_prefix.empty();
      ST_CHECK(st_conv_pack_weights(get(s.conv_prefix + ".weight"), s.has_bias ? get(s.conv_prefix + ".bias") : nullptr,
                                    bn ? get(s.bn_prefix + ".weight") : nullptr, bn ? get(s.bn_prefix + ".bias") : nullptr,
                                    bn ? get(s.bn_prefix + ".running_mean") : nullptr,
                                    bn ? get(s.bn_prefix + ".running_var") : nullptr, det->cfg.bn_eps, s.cout, pc.cin,
                                    pc.k, pc.k, wtmp.data(), btmp.data()));
      std::memcpy(host.data() + pc.wgt_off + (size_t)row * Kpad, wtmp.data(), sizeof(float) * (size_t)s.cout * Kpad);
      std::memcpy(host.data() + pc.bias_off + row, btmp.data(), sizeof(float) * (size_t)s.cout);
      row += s.cout;
    }
    if (pc.wino) ST_CHECK(wino_pack_weights(host.data() + pc.wgt_off, pc.cout, pc.cin, host.data() + pc.wino_off));
    if (pc.frag) ST_CHECK(front_pack_frags(host.data() + pc.wgt_off, pc.cout, pc.cin, host.data() + pc.frag_off));
    if (pc.tail) ST_CHECK(csp_tail_pack_frags(host.data() + pc.wgt_off, host.data() + pc.tail_off));
  }
  if (!det->wgt_dev) ST_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&det->wgt_dev), det->wgt_floats * sizeof(float)));
  ST_CHECK_HIP(hipMemcpy(det->wgt_dev, host.data(), det->wgt_floats * sizeof(float), hipMemcpyHostToDevice));
  det->finalized = true;
  return ST_OK;
}

extern "C" size_t st_detector_workspace_bytes(const StDetector* det) { return det ? det->ws_floats * sizeof(float) : 0; }
extern "C" size_t st_detector_head_floats(const StDetector* det) { return det ? det->head_floats : 0; }
extern "C" int st_detector_num_levels(const StDetector* det) { return det ? det->n_levels : 0; }
extern "C" double st_detector_macs(const StDetector* det) { return det ? det->macs : 0.0; }

extern "C" int st_detector_level_info(const StDetector* det, int level, int* h, int* w, int* stride,
                                      size_t* float_offset) {
  if (!det || level < 0 || level >= det->n_levels) return set_error(ST_ERR_INVALID, "st_detector_level_info: bad level");
  if (h) *h = det->lvl_h[level];
  if (w) *w = det->lvl_w[level];
  if (stride) *stride = det->lvl_stride[level];
  if (float_offset) *float_offset = det->lvl_off[level];
  return ST_OK;
}

namespace {

float* resolve(const StDetector* det, const TRef& t, float* ws, float* head, int img0 = 0) {
  if (t.buf == BUF_NONE) return nullptr;
  if (t.buf == BUF_HEAD) return head + t.base;
  // `adv` tensors are sub-batch windows of a full-batch buffer: move to image img0
  const size_t shift = t.adv ? (size_t)img0 * t.H * t.W * t.ld : 0;
  return ws + det->buf_off[t.buf] + t.base + shift;
}

StConvDesc conv_desc(const StDetector* det, const Op& o, int img0, float* ws, float* head) {
  const PackedConv& pc = det->convs[o.pc];
  StConvDesc d{};
  d.in_dev = resolve(det, o.in, ws, head, img0);
  d.N = o.in.N; d.Hi = o.in.H; d.Wi = o.in.W; d.Cin = pc.cin; d.in_ld = o.in.ld; d.in_off = o.in.off;
  d.wgt_dev = det->wgt_dev + pc.wgt_off;
  d.bias_dev = det->wgt_dev + pc.bias_off;
  d.Cout = pc.cout; d.KH = pc.k; d.KW = pc.k; d.stride = o.stride; d.pad = o.pad;
  d.out1_dev = resolve(det, o.out1, ws, head, img0); d.out1_ld = o.out1.ld; d.out1_off = o.out1.off;
  d.split = o.split;
  d.out2_dev = resolve(det, o.out2, ws, head, img0); d.out2_ld = o.out2.ld; d.out2_off = o.out2.off;
  d.up_dev = resolve(det, o.up, ws, head, img0); d.up_ld = o.up.ld; d.up_off = o.up.off;
  d.res_dev = resolve(det, o.res, ws, head, img0); d.res_ld = o.res.ld; d.res_off = o.res.off;
  d.post_scale = o.post_scale; d.act = o.act;
  d.wgt_wino_dev = pc.wino ? det->wgt_dev + pc.wino_off : nullptr;
  return d;
}

int launch_op(StDetector* det, Op& o, int img0, const float* const inputs[3], float* ws, float* head,
              hipStream_t stream) {
  switch (o.type) {
    case Op::FOCUS: {
      const float* src = inputs[o.focus_input];
      ST_REQUIRE(src != nullptr, "detector: input %d not provided (raw uint8 frames need the fused stem)", o.focus_input);
      float* dst = resolve(det, o.out1, ws, head) + (size_t)o.focus_batch_off * o.out1.H * o.out1.W * o.out1.ld;
      return focus_pack_launch(src, det->cfg.batch, 3, det->cfg.height, det->cfg.width, dst, stream);
    }
    case Op::STEM: {
      const float* src = inputs[o.focus_input];
      const StemRawInput* raw = det->raw_inputs[o.focus_input].frames ? &det->raw_inputs[o.focus_input] : nullptr;
      ST_REQUIRE(src != nullptr || raw != nullptr, "detector: input %d not provided", o.focus_input);
      const PackedConv& pc = det->convs[o.pc];
      float* dst = resolve(det, o.out1, ws, head) + (size_t)o.focus_batch_off * o.out1.H * o.out1.W * o.out1.ld;
      return stem_focus_conv_launch(src, det->cfg.batch, det->cfg.height, det->cfg.width, pc.stem_planes,
                                    det->wgt_dev + pc.wgt_off, det->wgt_dev + pc.bias_off, pc.cout, dst, o.out1.ld,
                                    o.out1.off, 1, stream, raw);
    }
    case Op::SPP: {
      float* x = resolve(det, o.in, ws, head, img0);
      float* out = resolve(det, o.out1, ws, head, img0);
      return spp_pool_launch(x, o.in.ld, o.in.off, o.in.N, o.in.H, o.in.W, o.in.C, out, o.out1.ld, o.out1.off,
                             stream);
    }
    case Op::PRED: {
      HeadPredArgs a{};
      for (int l = 0; l < 3; ++l) {
        const PackedConv &pcc = det->convs[o.pred_pcc[l]], &pcr = det->convs[o.pred_pcr[l]];
        HeadPredLevel& L = a.lv[l];
        L.cls = resolve(det, o.pred_cls[l], ws, head, img0); L.cls_ld = o.pred_cls[l].ld; L.cls_off = o.pred_cls[l].off;
        L.reg = resolve(det, o.pred_reg[l], ws, head, img0); L.reg_ld = o.pred_reg[l].ld; L.reg_off = o.pred_reg[l].off;
        L.wc = det->wgt_dev + pcc.wgt_off; L.bc = det->wgt_dev + pcc.bias_off;
        L.wr = det->wgt_dev + pcr.wgt_off; L.br = det->wgt_dev + pcr.bias_off;
        L.out = resolve(det, o.pred_out[l], ws, head, img0);
        L.M = o.pred_out[l].N * o.pred_out[l].H * o.pred_out[l].W;
        a.Kpad = round_up(pcc.cin, 32);
        a.nc = pcc.cout;
      }
      return head_pred_launch(a, det->convs[o.pred_pcc[0]].cin, stream);
    }
    case Op::CONV: {
      const PackedConv& pc = det->convs[o.pc];
      (void)pc;
      const StConvDesc d = conv_desc(det, o, img0, ws, head);
      return conv2d_launch(d, stream, det->force_variant >= 0 ? det->force_variant : o.tuned, &o.variant);
    }
  }
  return ST_OK;
}

// Runs the ops of phases [phase_lo, phase_hi].  Consecutive ops of one sub-batch group are executed
// sub-batch-major (all ops on images [0,SB), then [SB,2SB), ...) so their SB-sized intermediates are
// produced and consumed while still resident in the Infinity Cache.
int run_ops(StDetector* det, int phase_lo, int phase_hi, const float* const inputs[3], float* ws,
            float* head, hipStream_t stream) {
  const size_t ev_per_op = 2 * (size_t)det->max_group_count;
  if (det->timing && det->events.size() != ev_per_op * det->ops.size()) {
    for (auto& e : det->events) (void)hipEventDestroy(e);
    det->events.resize(ev_per_op * det->ops.size());
    for (auto& e : det->events) ST_CHECK_HIP(hipEventCreate(&e));
  }
  const size_t nops = det->ops.size();
  size_t oi = 0;
  while (oi < nops) {
    Op& first = det->ops[oi];
    if (first.phase < phase_lo || first.phase > phase_hi) { ++oi; continue; }
    size_t oe = oi + 1;
    if (first.group == 0 && first.wgroup > 0 && det->allow_wgroup && det->force_variant < 0) {
      // a run of independent Winograd layers (one tower depth of the head over the three levels): ONE grouped launch
      size_t we = oi;
      StConvDesc wd[8];
      bool ok = true;
      while (we < nops && we - oi < 8 && det->ops[we].wgroup == first.wgroup && det->ops[we].phase == first.phase &&
             det->ops[we].group == 0) {
        const Op& o = det->ops[we];
        ok = ok && o.type == Op::CONV && (o.tuned == 43 || o.tuned == 44);
        if (ok) wd[we - oi] = conv_desc(det, o, 0, ws, head);
        ++we;
      }
      const int nw = (int)(we - oi);
#ifdef ST_ABLATION
      for (size_t k = oi; k < we; ++k) ok = ok && !(k < det->skip.size() && det->skip[k]);
#endif
      if (ok && nw >= 2 && wino_group_applicable(wd, nw)) {
        if (det->timing) ST_CHECK_HIP(hipEventRecord(det->events[ev_per_op * oi], stream));
        ST_CHECK(wino_group_launch(wd, nw, stream));
        for (size_t k = oi; k < we; ++k) {   // the first op carries the launch's duration, the riders ~0
          if (det->timing && k > oi) ST_CHECK_HIP(hipEventRecord(det->events[ev_per_op * k], stream));
          if (det->timing) ST_CHECK_HIP(hipEventRecord(det->events[ev_per_op * k + 1], stream));
          det->ops[k].variant = k == oi ? 48 : 49;
        }
        oi = we;
        continue;
      }
    }
    if (first.group > 0)
      while (oe < nops && det->ops[oe].group == first.group && det->ops[oe].phase == first.phase) ++oe;
    else if (first.chain_next && oe < nops && det->ops[oe].group == 0 && det->ops[oe].phase == first.phase)
      ++oe;   // a chainable pair outside the sub-batch groups is still launched as one
    const StDetector::Group g = det->groups[first.group];
    for (int sbi = 0; sbi < g.count; ++sbi) {
      int fused_left = 0, fused_variant = 0;   // ops already computed by a preceding fused launch
      for (size_t k = oi; k < oe; ++k) {
        Op& o = det->ops[k];
        if (det->timing) ST_CHECK_HIP(hipEventRecord(det->events[ev_per_op * k + 2 * sbi], stream));
#ifdef ST_ABLATION   // tools-only build, timing only (results are garbage): st_detector_set_skip drops launches to
        const bool skipped = k < det->skip.size() && det->skip[k];   // bound what a faster kernel could buy
#else
        constexpr bool skipped = false;
#endif
        bool chained = false;
        if (!skipped && fused_left > 0) {   // this op was computed by a previous (chained / fused) launch
          --fused_left;
          o.variant = fused_variant;
          chained = true;
        } else if (!skipped && o.type == Op::CONV && o.front_next2 && k + 2 < oe && det->force_variant < 0 &&
                   det->allow_front && det->convs[det->ops[k + 1].pc].frag && det->convs[det->ops[k + 2].pc].frag) {
          const StConvDesc da = conv_desc(det, o, sbi * g.sb, ws, head);
          const StConvDesc dm = conv_desc(det, det->ops[k + 1], sbi * g.sb, ws, head);
          const StConvDesc dc = conv_desc(det, det->ops[k + 2], sbi * g.sb, ws, head);
          if (front_fused_applicable(da, dm, dc)) {
            ST_CHECK(front_fused_launch(da, dm, dc, det->wgt_dev + det->convs[det->ops[k + 1].pc].frag_off,
                                        det->wgt_dev + det->convs[det->ops[k + 2].pc].frag_off, stream));
            o.variant = fused_variant = 45;
            fused_left = 2;
            chained = true;
          }
        }
        if (!chained && !skipped && o.type == Op::CONV && o.tail_next && k + 1 < oe && det->force_variant < 0 &&
            det->allow_tail && det->convs[det->ops[k + 1].pc].tail) {
          const StConvDesc d2 = conv_desc(det, o, sbi * g.sb, ws, head);
          const StConvDesc df = conv_desc(det, det->ops[k + 1], sbi * g.sb, ws, head);
          if (csp_tail_applicable(d2, df)) {
            ST_CHECK(csp_tail_launch(d2, df, det->wgt_dev + det->convs[det->ops[k + 1].pc].tail_off, stream));
            o.variant = fused_variant = 56;
            fused_left = 1;
            chained = true;
          }
        }
        if (!chained && !skipped && o.type == Op::CONV && o.chain_next && k + 1 < oe && det->force_variant < 0 &&
            (o.tuned == 41 || o.tuned == 46) && det->ops[k + 1].tuned == o.tuned && det->allow_chain) {
          const StConvDesc da = conv_desc(det, o, sbi * g.sb, ws, head);
          const StConvDesc db = conv_desc(det, det->ops[k + 1], sbi * g.sb, ws, head);
          if (o.tuned == 41 ? pw_chain_applicable(da, db) : pwr_chain_applicable(da, db)) {
            ST_CHECK(o.tuned == 41 ? pw_conv_launch(da, stream, &db) : pwr_conv_launch(da, stream, &db));
            o.variant = fused_variant = o.tuned;
            fused_left = 1;
            chained = true;
          }
        }
        if (!skipped && !chained) ST_CHECK(launch_op(det, o, sbi * g.sb, inputs, ws, head, stream));
        if (det->timing) ST_CHECK_HIP(hipEventRecord(det->events[ev_per_op * k + 2 * sbi + 1], stream));
      }
    }
    oi = oe;
  }
  return ST_OK;
}

}  // namespace

extern "C" int st_detector_forward(StDetector* det, const float* img_dev, const float* disp_dev,
                                   void* workspace_dev, size_t workspace_bytes, st_stream_t stream,
                                   float* head_out_dev) {
  if (!det) return set_error(ST_ERR_INVALID, "st_detector_forward: null detector");
  if (!det->finalized) return set_error(ST_ERR_STATE, "st_detector_forward: call st_detector_finalize first");
  ST_REQUIRE(img_dev && (disp_dev || det->cfg.rgb_only) && workspace_dev && head_out_dev, "st_detector_forward: null pointer");
  ST_REQUIRE(!det->cfg.with_right_branch, "st_detector_forward: detector was built for stereo; use st_detector_forward_phase");
  if (workspace_bytes < det->ws_floats * sizeof(float))
    return set_error(ST_ERR_WORKSPACE, "st_detector_forward: workspace %zu < required %zu", workspace_bytes,
                     det->ws_floats * sizeof(float));
  const float* inputs[3] = {img_dev, disp_dev, nullptr};
  return run_ops(det, 0, 1, inputs, static_cast<float*>(workspace_dev), head_out_dev,
                 static_cast<hipStream_t>(stream));
}

// Phase API for the stereo configuration: phase 0 = stem+stage1 features of left (and right),
// phase 1 = disparity branch + fused trunk + neck + head.  Between the two the caller runs the
// cost-volume module on the "stage1_rgb" tap to produce disp_postp.
extern "C" int st_detector_forward_phase(StDetector* det, int phase, const float* img_dev,
                                         const float* disp_dev, const float* right_dev,
                                         void* workspace_dev, size_t workspace_bytes,
                                         st_stream_t stream, float* head_out_dev) {
  if (!det) return set_error(ST_ERR_INVALID, "st_detector_forward_phase: null detector");
  if (!det->finalized) return set_error(ST_ERR_STATE, "st_detector_forward_phase: call st_detector_finalize first");
  ST_REQUIRE(phase == 0 || phase == 1, "st_detector_forward_phase: phase must be 0 or 1");
  ST_REQUIRE(workspace_dev != nullptr, "st_detector_forward_phase: null workspace");
  if (workspace_bytes < det->ws_floats * sizeof(float))
    return set_error(ST_ERR_WORKSPACE, "st_detector_forward_phase: workspace %zu < required %zu", workspace_bytes,
                     det->ws_floats * sizeof(float));
  if (phase == 0) {
    ST_REQUIRE(img_dev != nullptr, "st_detector_forward_phase: null img");
    ST_REQUIRE(!det->cfg.with_right_branch || right_dev != nullptr, "st_detector_forward_phase: right image required");
  } else {
    ST_REQUIRE((disp_dev || det->cfg.rgb_only) && head_out_dev, "st_detector_forward_phase: null disp/head pointer");
  }
  const float* inputs[3] = {img_dev, disp_dev, right_dev};
  return run_ops(det, phase, phase, inputs, static_cast<float*>(workspace_dev), head_out_dev,
                 static_cast<hipStream_t>(stream));
}

// Phase 0 of the stereo configuration from RAW frames: the left (and right) images are N separate uint8 [3][h][w]
// device frames; the fused stem converts and pads them (to height x width, with pad_value) while it stages its input
// windows, so the cast + pad of the data preprocessor costs no pass over HBM and no fp32 copy of the images exists.
extern "C" int st_detector_forward_phase0_raw(StDetector* det, const unsigned char* const* left_frames_host,
                                              const unsigned char* const* right_frames_host, int h, int w,
                                              float pad_value, void* workspace_dev, size_t workspace_bytes,
                                              st_stream_t stream) {
  if (!det) return set_error(ST_ERR_INVALID, "st_detector_forward_phase0_raw: null detector");
  if (!det->finalized) return set_error(ST_ERR_STATE, "st_detector_forward_phase0_raw: call st_detector_finalize first");
  ST_REQUIRE(workspace_dev != nullptr && left_frames_host != nullptr, "st_detector_forward_phase0_raw: null pointer");
  ST_REQUIRE(!det->cfg.with_right_branch || right_frames_host != nullptr,
             "st_detector_forward_phase0_raw: right frames required");
  if (workspace_bytes < det->ws_floats * sizeof(float))
    return set_error(ST_ERR_WORKSPACE, "st_detector_forward_phase0_raw: workspace %zu < required %zu", workspace_bytes,
                     det->ws_floats * sizeof(float));
  det->raw_inputs[0] = StemRawInput{left_frames_host, h, w, pad_value};
  det->raw_inputs[2] = StemRawInput{right_frames_host, h, w, pad_value};
  const float* inputs[3] = {nullptr, nullptr, nullptr};
  const int rc = run_ops(det, 0, 0, inputs, static_cast<float*>(workspace_dev), nullptr, static_cast<hipStream_t>(stream));
  det->raw_inputs[0] = det->raw_inputs[2] = StemRawInput{nullptr, 0, 0, 0.f};
  return rc;
}

// The whole forward of the disparity-INPUT configuration (the reference's shipped one: precomputed disparity maps) with
// the image as RAW uint8 frames: as st_detector_forward, the RGB stem reading the frames itself.
extern "C" int st_detector_forward_raw(StDetector* det, const unsigned char* const* img_frames_host, int h, int w,
                                       float pad_value, const float* disp_dev, void* workspace_dev,
                                       size_t workspace_bytes, st_stream_t stream, float* head_out_dev) {
  if (!det) return set_error(ST_ERR_INVALID, "st_detector_forward_raw: null detector");
  if (!det->finalized) return set_error(ST_ERR_STATE, "st_detector_forward_raw: call st_detector_finalize first");
  ST_REQUIRE(img_frames_host && (disp_dev || det->cfg.rgb_only) && workspace_dev && head_out_dev, "st_detector_forward_raw: null pointer");
  ST_REQUIRE(!det->cfg.with_right_branch, "st_detector_forward_raw: detector was built for stereo; use st_detector_forward_phase0_raw");
  if (workspace_bytes < det->ws_floats * sizeof(float))
    return set_error(ST_ERR_WORKSPACE, "st_detector_forward_raw: workspace %zu < required %zu", workspace_bytes,
                     det->ws_floats * sizeof(float));
  det->raw_inputs[0] = StemRawInput{img_frames_host, h, w, pad_value};
  const float* inputs[3] = {nullptr, disp_dev, nullptr};
  const int rc = run_ops(det, 0, 1, inputs, static_cast<float*>(workspace_dev), head_out_dev,
                         static_cast<hipStream_t>(stream));
  det->raw_inputs[0] = StemRawInput{nullptr, 0, 0, 0.f};
  return rc;
}

// Per-op timing for bench.py / profiling: when enabled, every op of the next forward is bracketed
// by hipEvents on the caller's stream.  st_detector_op_times synchronises on those events.
extern "C" int st_detector_set_timing(StDetector* det, int enable) {
  if (!det) return set_error(ST_ERR_INVALID, "st_detector_set_timing: null detector");
  det->timing = enable != 0;
  return ST_OK;
}

extern "C" int st_detector_num_ops(const StDetector* det) { return det ? (int)det->ops.size() : 0; }

#ifdef ST_ABLATION
// Tools-only build (make ABLATION=1): drop the launches of the listed ops (timing experiments; results are
// garbage) and/or forbid the chained 1x1 pair.  Not part of the product library.
extern "C" int st_detector_set_skip(StDetector* det, const int* ops, int n, int allow_chain) {
  if (!det) return set_error(ST_ERR_INVALID, "st_detector_set_skip: null detector");
  det->skip.assign(det->ops.size(), 0);
  for (int i = 0; i < n; ++i)
    if (ops[i] >= 0 && ops[i] < (int)det->ops.size()) det->skip[ops[i]] = 1;
  det->allow_chain = allow_chain != 0;
  return ST_OK;
}
#endif

// kind: 0 focus-pack, 1 conv, 2 spp; variant = conv tile variant (0..4) or -1; macs = conv MACs
extern "C" int st_detector_op_times(StDetector* det, int cap, float* ms, int* kind, int* variant, double* macs,
                                    int* phase) {
  if (!det) return set_error(ST_ERR_INVALID, "st_detector_op_times: null detector");
  const size_t ev_per_op = 2 * (size_t)det->max_group_count;
  ST_REQUIRE(det->events.size() == ev_per_op * det->ops.size(), "st_detector_op_times: no timed forward has run");
  ST_REQUIRE(cap >= (int)det->ops.size(), "st_detector_op_times: capacity %d < %zu ops", cap, det->ops.size());
  for (size_t i = 0; i < det->ops.size(); ++i) {
    float t = 0.f;
    const int cnt = det->groups[det->ops[i].group].count;
    for (int sbi = 0; sbi < cnt; ++sbi) {
      ST_CHECK_HIP(hipEventSynchronize(det->events[ev_per_op * i + 2 * sbi + 1]));
      float dt = 0.f;
      ST_CHECK_HIP(hipEventElapsedTime(&dt, det->events[ev_per_op * i + 2 * sbi], det->events[ev_per_op * i + 2 * sbi + 1]));
      t += dt;
    }
    if (ms) ms[i] = t;
    if (kind) kind[i] = det->ops[i].type == Op::FOCUS ? 0 : det->ops[i].type == Op::SPP ? 2 : 1;  // STEM, PRED count as conv
    if (variant) variant[i] = det->ops[i].variant;
    if (macs) macs[i] = det->ops[i].macs;
    if (phase) phase[i] = det->ops[i].phase;
  }
  return ST_OK;
}

// Measure every valid tile variant of every conv op on the real shapes (HIP events on `stream`,
// workspace contents are whatever the last forward left) and keep the fastest.  Host-synchronous;
// call once after st_detector_finalize, outside any timed region.
extern "C" int st_detector_autotune(StDetector* det, void* workspace_dev, size_t workspace_bytes,
                                    float* head_out_dev, st_stream_t stream_, int reps) {
  if (!det) return set_error(ST_ERR_INVALID, "st_detector_autotune: null detector");
  if (!det->finalized) return set_error(ST_ERR_STATE, "st_detector_autotune: call st_detector_finalize first");
  ST_REQUIRE(workspace_dev && head_out_dev, "st_detector_autotune: null pointer");
  if (workspace_bytes < det->ws_floats * sizeof(float))
    return set_error(ST_ERR_WORKSPACE, "st_detector_autotune: workspace too small");
  if (reps <= 0) reps = 5;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  hipEvent_t e0, e1;
  ST_CHECK_HIP(hipEventCreate(&e0));
  ST_CHECK_HIP(hipEventCreate(&e1));
  const float* inputs[3] = {nullptr, nullptr, nullptr};
  const bool was_timing = det->timing;
  det->timing = false;
  int rc = ST_OK;
  std::vector<Op> saved = det->ops;
  for (size_t oi = 0; oi < saved.size() && rc == ST_OK; ++oi) {
    if (saved[oi].type != Op::CONV) continue;
    // run only this op: temporarily make it the sole op of a private phase
    det->ops.assign(1, saved[oi]);
    det->ops[0].phase = 0;
    float best = 1e30f;
    int best_v = -1;
    // candidates: every staged variant; the wave-specialised ones (>= 22) measured slower on every layer
    // shape of this network and are left out of the search
    // + variant 41, the streaming 1x1 kernel (pointwise_conv.hip), where the layer shape allows it
    const Op& so = saved[oi];
    StConvDesc probe{};
    probe.in_dev = static_cast<float*>(workspace_dev);   // alignment / size checks only
    probe.N = so.in.N; probe.Hi = so.in.H; probe.Wi = so.in.W; probe.Cin = det->convs[so.pc].cin;
    probe.in_ld = so.in.ld; probe.in_off = so.in.off; probe.Cout = det->convs[so.pc].cout;
    probe.KH = probe.KW = det->convs[so.pc].k; probe.stride = so.stride; probe.pad = so.pad;
    probe.out1_ld = so.out1.ld;
    probe.out2_dev = so.out2.valid() ? static_cast<float*>(workspace_dev) : nullptr; probe.out2_ld = so.out2.ld;
    probe.res_dev = so.res.valid() ? probe.in_dev : nullptr; probe.res_ld = so.res.ld;
    probe.up_dev = so.up.valid() ? static_cast<float*>(workspace_dev) : nullptr;
    probe.out1_dev = static_cast<float*>(workspace_dev);
    probe.out1_off = so.out1.off; probe.res_off = so.res.off;
    const bool pw_ok = pw_conv_applicable(probe);
    const bool dc_ok = dc_conv_applicable(probe);   // + variant 42, the direct 3x3 kernel (direct_conv.hip)
    probe.wgt_wino_dev = det->convs[so.pc].wino ? det->wgt_dev + det->convs[so.pc].wino_off : nullptr;
    const bool wn_ok = !det->no_wino && wino_conv_applicable(probe);   // + variant 43, Winograd F(2x2,3x3) (wino_conv.hip)
    const int ncand = std::min(conv_variant_count(), 22);
    const bool wn_narrow_ok = wn_ok && det->convs[so.pc].cout % 64 == 0;   // + variant 44: 32-cout Winograd workgroups
    probe.wgt_dev = det->wgt_dev + det->convs[so.pc].wgt_off;
    probe.bias_dev = det->wgt_dev + det->convs[so.pc].bias_off;
    probe.split = so.split; probe.out2_off = so.out2.off;
    const bool pr_ok = pwr_conv_applicable(probe);   // + variant 46: 1x1 with LDS-resident weights (pointwise_resident.hip)
    const int cout_pad_s = round_up(det->convs[so.pc].cout, 32);
    for (int vi = 0; vi <= ncand + 10 && rc == ST_OK; ++vi) {
      const int v = vi < ncand ? vi : vi > ncand + 4 ? 50 + (vi - ncand - 5) : vi == ncand + 4 ? 46 : 41 + (vi - ncand);
      if (v >= 50) {   // split-operand instances: only when the caller allowed them; tiles must divide Cout
        if (!((det->allow_split >> (v - 50)) & 1) || cout_pad_s % ((v - 50) % 3 == 0 ? 128 : 64) != 0) continue;
      } else if (v == 46 ? !pr_ok : v == 41 ? !pw_ok : v == 42 ? !dc_ok : v == 43 ? !wn_ok : v == 44 ? !wn_narrow_ok
                                                                  : !conv_variant_valid(v, det->convs[saved[oi].pc].cout))
        continue;
      det->force_variant = v;
      rc = run_ops(det, 0, 0, inputs, static_cast<float*>(workspace_dev), head_out_dev, stream);  // warm
      if (rc != ST_OK) break;
      // min over `reps` individually timed launches (a mean over a burst is too noisy to rank 20 variants)
      float ms = 1e30f;
      for (int r = 0; r < reps && rc == ST_OK; ++r) {
        if (hipEventRecord(e0, stream) != hipSuccess) { rc = set_error(ST_ERR_HIP, "autotune: event record"); break; }
        rc = run_ops(det, 0, 0, inputs, static_cast<float*>(workspace_dev), head_out_dev, stream);
        if (rc != ST_OK) break;
        if (hipEventRecord(e1, stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) {
          rc = set_error(ST_ERR_HIP, "autotune: event sync");
          break;
        }
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e0, e1);
        ms = std::min(ms, t);
      }
      if (rc != ST_OK) break;
      // hysteresis: a later candidate must win by 3 % (min-of-N timings of near-equal kernels are noise: the choice
      // should not flip between runs, and the tuned plan stays comparable from run to run)
      if (best_v < 0 || ms < best * 0.97f) { best = ms; best_v = v; }
    }
    saved[oi].tuned = best_v;
  }
  det->force_variant = -1;
  det->ops = saved;
  det->timing = was_timing;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

// Allow (1) / forbid (0, default) the split-operand (bf16x3) instances in st_detector_autotune's search.  The instances
// are PARKED in the tools-only build (round 5: they do not pass the frozen parity gate): the product library
// accepts 0 only.
extern "C" int st_split_instances_available(void) {
#ifdef ST_ABLATION
  return 1;
#else
  return 0;
#endif
}

extern "C" int st_detector_set_split(StDetector* det, int allow) {
  if (!det) return set_error(ST_ERR_INVALID, "st_detector_set_split: null detector");
  if (allow != 0 && !st_split_instances_available())
    return set_error(ST_ERR_INVALID, "st_detector_set_split: the split-operand (bf16x3) instances are not part of the product "
                                     "library (parked in the tools build: make ABLATION=1, ST_LIBRARY=<that library>)");
  det->allow_split = allow == 1 ? 0x3F : (allow & 0x3F);   // 1 = all six instances; otherwise a mask (bit i = variant 50 + i)
  return ST_OK;
}

// id 40 = the fused Focus+stem kernel (stem_focus_conv.hip), reported with the conv ops
// id 41 = the streaming 1x1 kernel for narrow layers (pointwise_conv.hip)
extern "C" const char* st_conv_variant_name(int id) {
  return id == 56 ? "wino32tail" : id == 47 ? "headpred" : id == 46 ? "pwres" : id == 45 ? "front3x3s2" : id == 40 ? "stem6x6s2" : id == 41 ? "pw128" : id == 42 ? "dc4x32" : id == 43 ? "wino2x2" : id == 44 ? "wino2x2n" : id == 48 ? "wino2x2g" : id == 49 ? "wino2x2g+" : id == 50 ? "split128x128" : id == 51 ? "split64x64" : id == 52 ? "split128x64" : id == 53 ? "split128x128k16" : id == 54 ? "split64x64k16" : id == 55 ? "split128x64k16" : conv_variant_name(id);
}
extern "C" const char* st_conv_variant_signature(int id) {
  return id == 56 ? "wino_csp_tail (bottleneck conv2 + final_conv, persistent)" : id == 47 ? "head_pred" : id == 46 ? "pw_resident" : id == 45 ? "front_s2_csp" : id == 40 ? "stem_focus_conv" : id == 41 ? "pw_conv" : id == 42 ? "direct_conv3x3" : id >= 50 && id <= 55 ? "conv_split (bf16x3 operands)" : id == 43 ? "wino_conv3x3" : id == 48 ? "wino_conv3x3 grouped launch" : id == 49 ? "wino_conv3x3 grouped launch (rider: computed by the preceding op's launch)" : id == 44 ? "wino_conv3x3 narrow"
                                                              : conv_variant_signature(id);
}

// Read / restore the per-op tile choice (one int per op, -1 = heuristic) so a tuning result can be
// cached across processes (e.g. to keep autotune launches out of a rocprofv3 trace).
extern "C" int st_detector_get_tuning(const StDetector* det, int* variants, int cap) {
  if (!det || !variants) return set_error(ST_ERR_INVALID, "st_detector_get_tuning: null argument");
  ST_REQUIRE(cap >= (int)det->ops.size(), "st_detector_get_tuning: capacity too small");
  for (size_t i = 0; i < det->ops.size(); ++i) variants[i] = det->ops[i].tuned;
  return ST_OK;
}

extern "C" int st_detector_set_tuning(StDetector* det, const int* variants, int n) {
  if (!det || !variants) return set_error(ST_ERR_INVALID, "st_detector_set_tuning: null argument");
  ST_REQUIRE(n == (int)det->ops.size(), "st_detector_set_tuning: expected %zu entries, got %d", det->ops.size(), n);
  for (int i = 0; i < n; ++i) {
    const Op& o = det->ops[i];
    if (variants[i] < 0 || o.type != Op::CONV) continue;
    if (variants[i] >= 41 && variants[i] <= 46) continue;   // own applicability checks run at launch
    if (variants[i] >= 50 && variants[i] <= 55) {           // split-operand instances: the tile must divide Cout
      ST_REQUIRE(round_up(det->convs[o.pc].cout, 32) % ((variants[i] - 50) % 3 == 0 ? 128 : 64) == 0,
                 "st_detector_set_tuning: split variant %d invalid for op %d", variants[i], i);
      continue;
    }
    ST_REQUIRE(conv_variant_valid(variants[i], det->convs[o.pc].cout), "st_detector_set_tuning: variant %d invalid for op %d",
               variants[i], i);
  }
  for (int i = 0; i < n; ++i)
    if (det->ops[i].type == Op::CONV) det->ops[i].tuned = variants[i];
  return ST_OK;
}

// Human-readable description of op i (profiling aid).
extern "C" int st_detector_op_desc(const StDetector* det, int i, char* buf, int cap) {
  if (!det || i < 0 || i >= (int)det->ops.size() || !buf || cap <= 0)
    return set_error(ST_ERR_INVALID, "st_detector_op_desc: bad argument");
  const Op& o = det->ops[i];
  if (o.type == Op::FOCUS) {
    snprintf(buf, (size_t)cap, "focus_pack input=%d", o.focus_input);
  } else if (o.type == Op::STEM) {
    snprintf(buf, (size_t)cap, "stem focus+conv6x6 s2 input=%d N=%d Hi=%d Wi=%d Cin=%d Cout=%d  %s", o.focus_input,
             det->cfg.batch, det->cfg.height, det->cfg.width, det->convs[o.pc].stem_planes, det->convs[o.pc].cout,
             det->convs[o.pc].srcs[0].conv_prefix.c_str());
  } else if (o.type == Op::SPP) {
    snprintf(buf, (size_t)cap, "spp_pool N=%d H=%d W=%d C=%d", o.in.N, o.in.H, o.in.W, o.in.C);
  } else if (o.type == Op::PRED) {
    snprintf(buf, (size_t)cap, "head conv_cls + conv_reg + conv_obj, 3 levels (%dx%d, %dx%d, %dx%d) N=%d Cin=%d  %s",
             o.pred_out[0].H, o.pred_out[0].W, o.pred_out[1].H, o.pred_out[1].W, o.pred_out[2].H, o.pred_out[2].W,
             o.pred_out[0].N, det->convs[o.pred_pcc[0]].cin, det->convs[o.pred_pcc[0]].srcs[0].conv_prefix.c_str());
  } else {
    const PackedConv& pc = det->convs[o.pc];
    snprintf(buf, (size_t)cap, "conv k%d s%d N=%dx%d Hi=%d Wi=%d Cin=%d Cout=%d%s%s%s  %s", pc.k, o.stride, o.in.N,
             det->groups[o.group].count, o.in.H, o.in.W, pc.cin, pc.cout, o.res.valid() ? " +res" : "",
             o.up.valid() ? " +up" : "", o.out2.valid() ? " +split" : "", pc.srcs[0].conv_prefix.c_str());
  }
  return ST_OK;
}

extern "C" int st_detector_tap(const StDetector* det, const char* name, const void* workspace_dev,
                               const float** ptr_dev, int* N, int* C, int* H, int* W, int* ld) {
  if (!det || !name) return set_error(ST_ERR_INVALID, "st_detector_tap: null argument");
  auto it = det->taps.find(name);
  if (it == det->taps.end()) return set_error(ST_ERR_NOTFOUND, "st_detector_tap: unknown tap '%s'", name);
  const TRef& t = it->second;
  if (ptr_dev) *ptr_dev = static_cast<const float*>(workspace_dev) + det->buf_off[t.buf] + t.off;
  if (N) *N = t.N;
  if (ld) *ld = t.ld;
  if (C) *C = t.C;
  if (H) *H = t.H;
  if (W) *W = t.W;
  return ST_OK;
}
