// Host routine of the dataset reader: PNG scanline un-filtering (RFC 2083 §6: None / Sub / Up / Average / Paeth).
//
// The reference decodes its uint8 images and uint16 disparity / depth maps with OpenCV behind mmcv.imfrombytes
// (mmtrack/datasets/transforms/loading_disparity.py:74-75, 213-215; mmcv / cv2 are absent here).  The container
// parsing and the inflate step are done with the Python standard library (zlib); the only part that is a serial
// per-byte loop - reversing the scanline filters - is this function.  No GPU involved.
#include <cstdint>
#include <cstdlib>

#include "st_common.h"

extern "C" int st_png_unfilter(const uint8_t* filtered, int height, int stride, int bpp, uint8_t* out) {
  using namespace st;
  ST_REQUIRE(filtered && out && height > 0 && stride > 0 && bpp > 0 && bpp <= 8 && stride % bpp == 0,
             "st_png_unfilter: bad arguments");
  const uint8_t* prev = nullptr;
  for (int y = 0; y < height; ++y) {
    const uint8_t* row = filtered + (size_t)y * (stride + 1);
    const int ft = row[0];
    const uint8_t* src = row + 1;
    uint8_t* dst = out + (size_t)y * stride;
    switch (ft) {
      case 0:
        for (int i = 0; i < stride; ++i) dst[i] = src[i];
        break;
      case 1:
        for (int i = 0; i < stride; ++i) dst[i] = (uint8_t)(src[i] + (i >= bpp ? dst[i - bpp] : 0));
        break;
      case 2:
        for (int i = 0; i < stride; ++i) dst[i] = (uint8_t)(src[i] + (prev ? prev[i] : 0));
        break;
      case 3:
        for (int i = 0; i < stride; ++i) {
          const int a = i >= bpp ? dst[i - bpp] : 0, b = prev ? prev[i] : 0;
          dst[i] = (uint8_t)(src[i] + ((a + b) >> 1));
        }
        break;
      case 4:
        for (int i = 0; i < stride; ++i) {
          const int a = i >= bpp ? dst[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
          const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
          const int pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          dst[i] = (uint8_t)(src[i] + pr);
        }
        break;
      default:
        return set_error(ST_ERR_INVALID, "st_png_unfilter: row %d has filter type %d (not a PNG filter)", y, ft);
    }
    prev = dst;
  }
  return ST_OK;
}
