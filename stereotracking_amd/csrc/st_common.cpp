// Error plumbing + host-side weight packing for libstereotrack_hip.
#include <atomic>

#include "st_common.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace st {

std::string& last_error() {
  static thread_local std::string s;
  return s;
}

int set_error(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

}  // namespace st

extern "C" int st_version(void) { return ST_VERSION; }
extern "C" int st_head_row_floats(int num_classes) { return st::head_row_floats(num_classes < 1 ? 1 : num_classes); }
extern "C" const char* st_last_error(void) { return st::last_error().c_str(); }

extern "C" size_t st_conv_packed_floats(int Cout, int Cin, int KH, int KW) {
  return (size_t)st::round_up(Cout, 32) * st::round_up(KH * KW * Cin, 32);
}

// Conv2d weight [Cout][Cin][KH][KW] (+BN running stats) -> [CoutPad][Kpad], k = (kh*KW+kw)*Cin+ci.
// BN folding is done in fp64 and rounded once to fp32:
//   w' = w * gamma / sqrt(var + eps),  b' = beta + (conv_bias - mean) * gamma / sqrt(var + eps)
extern "C" int st_conv_pack_weights(const float* w, const float* conv_bias, const float* bn_gamma,
                                    const float* bn_beta, const float* bn_mean,
                                    const float* bn_var, double bn_eps, int Cout, int Cin, int KH,
                                    int KW, float* wgt_out, float* bias_out) {
  if (!w || !wgt_out || !bias_out || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0)
    return st::set_error(ST_ERR_INVALID, "st_conv_pack_weights: bad argument");
  const bool has_bn = bn_gamma != nullptr;
  if (has_bn && (!bn_beta || !bn_mean || !bn_var))
    return st::set_error(ST_ERR_INVALID, "st_conv_pack_weights: incomplete BN parameters");
  const int K = KH * KW * Cin, Kpad = st::round_up(K, 32), CoutPad = st::round_up(Cout, 32);
  std::memset(wgt_out, 0, sizeof(float) * (size_t)CoutPad * Kpad);
  std::memset(bias_out, 0, sizeof(float) * (size_t)CoutPad);
  for (int co = 0; co < Cout; ++co) {
    double scale = 1.0, shift = conv_bias ? (double)conv_bias[co] : 0.0;
    if (has_bn) {
      scale = (double)bn_gamma[co] / std::sqrt((double)bn_var[co] + bn_eps);
      shift = (double)bn_beta[co] + (shift - (double)bn_mean[co]) * scale;
    }
    bias_out[co] = (float)shift;
    float* dst = wgt_out + (size_t)co * Kpad;
    for (int ci = 0; ci < Cin; ++ci)
      for (int kh = 0; kh < KH; ++kh)
        for (int kw = 0; kw < KW; ++kw) {
          const double v = (double)w[(((size_t)co * Cin + ci) * KH + kh) * KW + kw] * scale;
          dst[(kh * KW + kw) * Cin + ci] = (float)v;
        }
  }
  return ST_OK;
}
