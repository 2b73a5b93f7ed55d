// cvlite.h -- the few OpenCV value types the reference's hot-path signatures mention, for builds
// where <opencv2/core.hpp> is not available (this image has no OpenCV).  When ORBHIP_WITH_OPENCV is
// defined the real cv:: types are used instead and this header is skipped.  Layout-compatible with
// OpenCV's: cv::KeyPoint is 28 bytes {pt.x, pt.y, size, angle, response, octave, class_id}.
#pragma once
#ifndef ORBHIP_WITH_OPENCV
#include <cstdint>
#include <cstring>
#include <vector>
namespace cv {
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float a, float b) : x(a), y(b) {} };
struct KeyPoint {
    Point2f pt; float size, angle, response; int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");
enum { CV_8U = 0 };
// Minimal single-channel 8-bit matrix: owns or borrows a row-major buffer.
class Mat {
public:
    int rows, cols; size_t step; uint8_t *data;
    Mat() : rows(0), cols(0), step(0), data(nullptr) {}
    Mat(int r, int c, int /*type*/) { create(r, c, CV_8U); }
    Mat(int r, int c, int /*type*/, void *borrowed, size_t st = 0) : rows(r), cols(c), step(st ? st : (size_t)c), data((uint8_t *)borrowed) {}
    void create(int r, int c, int /*type*/) { rows = r; cols = c; step = (size_t)c; store.assign((size_t)r * c, 0); data = store.data(); }
    void release() { rows = cols = 0; step = 0; data = nullptr; store.clear(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    uint8_t *ptr(int r) { return data + (size_t)r * step; }
    const uint8_t *ptr(int r) const { return data + (size_t)r * step; }
    template <typename T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + (size_t)r * step); }
    Mat row(int r) const { return Mat(1, cols, CV_8U, data + (size_t)r * step, step); }
private:
    std::vector<uint8_t> store;
};
typedef const Mat &InputArray;
typedef Mat &OutputArray;
}  // namespace cv
#endif
