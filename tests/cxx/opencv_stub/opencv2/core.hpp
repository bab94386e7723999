// tests/cxx/opencv_stub/opencv2/core.hpp -- NOT OpenCV.  A dozen declarations with the names and signatures the
// WSAMD_WITH_OPENCV adapters of stereo_reconstruction_amd/host/window_search.hpp use (cv::Mat: type(), data, rows, cols,
// step, Mat(rows, cols, type), at<T>(y, x); CV_8UC3, CV_64F, CV_Assert), so that the adapter block -- which this image,
// having no OpenCV, could never compile -- at least goes through a compiler and runs once
// (tests/test_io_and_boundary.py::test_opencv_adapters_compile_against_a_stub).  It proves the adapters' syntax and
// types against THIS declaration of the interface, not against OpenCV itself; INTEGRATION.md says so.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>

#define CV_8UC3 16
#define CV_64F 6
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error("CV_Assert: " #expr); } while (0)

namespace cv {
struct MatStep {
    size_t v = 0;
    operator size_t() const { return v; }
};
class Mat {
public:
    Mat() = default;
    Mat(int r, int c, int t) : rows(r), cols(c), type_(t)
    {
        const size_t esz = t == CV_64F ? 8 : t == CV_8UC3 ? 3 : 1;
        step.v = esz * (size_t)c;
        owner_.reset(static_cast<uint8_t *>(std::calloc((size_t)r * step.v + 1, 1)), std::free);
        data = owner_.get();
    }
    int type() const { return type_; }
    template <typename T> T &at(int y, int x) { return *reinterpret_cast<T *>(data + (size_t)y * step.v + sizeof(T) * (size_t)x); }
    template <typename T> const T &at(int y, int x) const { return *reinterpret_cast<const T *>(data + (size_t)y * step.v + sizeof(T) * (size_t)x); }
    uint8_t *data = nullptr;
    int rows = 0, cols = 0;
    MatStep step;

private:
    int type_ = 0;
    std::shared_ptr<uint8_t> owner_;
};
} // namespace cv
