// cvlite.h -- the few OpenCV value types the reference's hot-path signatures mention, for builds
// where <opencv2/core.hpp> is not available (this image has no OpenCV).  When ORBHIP_WITH_OPENCV is
// defined the real cv:: types are used instead and this header is skipped.  Layout-compatible with
// OpenCV's: cv::KeyPoint is 28 bytes {pt.x, pt.y, size, angle, response, octave, class_id}.
#pragma once
#ifndef ORBHIP_WITH_OPENCV
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
namespace cv {
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float a, float b) : x(a), y(b) {} };
struct Point3f { float x, y, z; Point3f() : x(0), y(0), z(0) {} Point3f(float a, float b, float c) : x(a), y(b), z(c) {} };
struct KeyPoint {
    Point2f pt; float size, angle, response; int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");
}  // namespace cv
// OpenCV's type codes are macros (CV_8U == 0, CV_32F == 5): code written against the reference uses the bare names
#ifndef CV_8U
#define CV_8U 0
#define CV_32F 5
#endif
namespace cv {
// Minimal single-channel matrix (8-bit or float): owns (shared, like cv::Mat's refcount) or borrows a row-major buffer.
class Mat {
public:
    int rows, cols; size_t step; uint8_t *data;
    Mat() : rows(0), cols(0), step(0), data(nullptr), type_(CV_8U) {}
    Mat(int r, int c, int type) : type_(CV_8U) { create(r, c, type); }
    Mat(int r, int c, int type, void *borrowed, size_t st = 0)
        : rows(r), cols(c), step(st ? st : (size_t)c * esz(type)), data((uint8_t *)borrowed), type_(type) {}
    void create(int r, int c, int type)
    {
        if (r == rows && c == cols && type == type_ && store && data == store->data()) return;     // as cv::Mat::create: keeps a fitting buffer
        rows = r; cols = c; type_ = type; step = (size_t)c * esz(type);
        store = std::make_shared<std::vector<uint8_t>>((size_t)r * step, 0); data = store->data();
    }
    void release() { rows = cols = 0; step = 0; data = nullptr; store.reset(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    size_t elemSize() const { return esz(type_); }
    uint8_t *ptr(int r) { return data + (ptrdiff_t)r * (ptrdiff_t)step; }            // r may be negative inside a padded parent (ROI views)
    const uint8_t *ptr(int r) const { return data + (ptrdiff_t)r * (ptrdiff_t)step; }
    template <typename T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + (size_t)r * step); }
    template <typename T> T *ptr(int r = 0) { return reinterpret_cast<T *>(data + (size_t)r * step); }
    template <typename T> T &at(int r, int c) { return reinterpret_cast<T *>(data + (size_t)r * step)[c]; }
    template <typename T> const T &at(int r, int c) const { return reinterpret_cast<const T *>(data + (size_t)r * step)[c]; }
    template <typename T> T &at(int i) { return cols == 1 ? at<T>(i, 0) : at<T>(0, i); }                 // vectors, as cv::Mat::at(int)
    template <typename T> const T &at(int i) const { return cols == 1 ? at<T>(i, 0) : at<T>(0, i); }
    Mat row(int r) const { Mat m(1, cols, type_, data + (size_t)r * step, step); m.store = store; return m; }
    Mat clone() const
    {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; r++) memcpy(m.ptr(r), ptr(r), (size_t)cols * esz(type_));
        return m;
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    static Mat eye(int r, int c, int type)
    {
        Mat m(r, c, type);
        for (int i = 0; i < r && i < c; i++) { if (type == CV_32F) m.at<float>(i, i) = 1.f; else m.at<uint8_t>(i, i) = 1; }
        return m;
    }
private:
    static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
    std::shared_ptr<std::vector<uint8_t>> store;
    int type_;
};
typedef const Mat &InputArray;
typedef Mat &OutputArray;
}  // namespace cv
#endif
