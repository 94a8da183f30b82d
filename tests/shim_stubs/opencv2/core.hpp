// Minimal stand-in for <opencv2/core.hpp>: declarations only, for -fsyntax-only checks of integration/reference_shim/
// (see tests/shim_stubs/README.md).  cv::Mat keeps OpenCV's documented header semantics that matter to the shims:
// row(i) returns a header sharing the parent's buffer, whose datastart / dataend still span the WHOLE parent
// (that is why the shims use ptr<uint8_t>(r), never datastart, for a row).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#define CV_8U 0
#define CV_32F 5

namespace cv {

typedef unsigned char uchar;
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
template <typename T, int N> struct Vec { T val[N]; T& operator[](int i); const T& operator[](int i) const; };
using Vec3b = Vec<uchar, 3>;
enum NormTypes { NORM_INF = 1, NORM_L1 = 2, NORM_L2 = 4, NORM_L2SQR = 5, NORM_HAMMING = 6, NORM_HAMMING2 = 7 };

class Mat {
  public:
    Mat();
    int rows, cols;
    uchar* data;
    const uchar* datastart;
    const uchar* dataend;
    int type() const;
    int depth() const;
    int channels() const;
    bool empty() const;
    bool isContinuous() const;
    size_t elemSize() const;
    size_t total() const;
    Mat row(int r) const;
    Mat clone() const;
    template <typename T> T* ptr(int r = 0);
    template <typename T> const T* ptr(int r = 0) const;
    template <typename T> T& at(int r, int c);
    template <typename T> const T& at(int r, int c) const;
};

}  // namespace cv
