// window_search.hpp -- C++ facade over the C-ABI (include/ws_stereo.h) with the reference's
// class and method names, so that the call sites of the reference keep reading the same:
//
//   reference (src/WindowSearch/BlockSearch.h:9-46, LinearSearch.h:9-20,
//              src/Rectification/rectification.hpp:50-51,64-66, rectification.cpp:66-88)
//       auto bs  = BlockSearch(leftRectified, rightRectified, blockSize, minD, maxD);
//       cv::Mat d = bs.computeDisparityMapLeft(smoothFactor);            // CV_64F
//   here
//       auto bs  = wsamd::BlockSearch(wsamd::view(left), wsamd::view(right), blockSize, minD, maxD);
//       wsamd::MatF64 d = bs.computeDisparityMapLeft(smoothFactor);      // doubles, row-major
//
// Header only; link against libws_stereo.so.  Errors the reference raises as cv::Exception
// (even blockSize, ROI outside the image) and everything the device cannot run surface as
// wsamd::Error.  Define WSAMD_WITH_OPENCV before including to get cv::Mat adapters.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ws_stereo.h"

#ifdef WSAMD_WITH_OPENCV
#include <opencv2/core.hpp>
#endif

namespace wsamd {

class Error : public std::runtime_error {
public:
    Error(int code, const std::string &what) : std::runtime_error(what), code_(code) {}
    int code() const { return code_; }

private:
    int code_;
};

// A CV_8UC3 image header on caller-owned pixels (what a cv::Mat header is to BlockSearch).
struct Image8UC3 {
    const uint8_t *data = nullptr;
    int rows = 0, cols = 0;
    size_t step = 0; // bytes per row
};

inline Image8UC3 view(const uint8_t *bgr, int rows, int cols, size_t step = 0)
{
    Image8UC3 v;
    v.data = bgr;
    v.rows = rows;
    v.cols = cols;
    v.step = step ? step : static_cast<size_t>(cols) * 3;
    return v;
}

// A CV_64F map owned by value, as the reference's methods return it (BlockSearch.cpp:33,85).
class MatF64 {
public:
    MatF64() = default;
    MatF64(int rows, int cols) : rows(rows), cols(cols), buf_(static_cast<size_t>(rows) * cols, 0.0) {}
    int rows = 0, cols = 0;
    double &at(int y, int x) { return buf_[static_cast<size_t>(y) * cols + x]; }
    double at(int y, int x) const { return buf_[static_cast<size_t>(y) * cols + x]; }
    double *ptr() { return buf_.data(); }
    const double *ptr() const { return buf_.data(); }
    bool empty() const { return buf_.empty(); }

private:
    std::vector<double> buf_;
};

// One ws_context per device, shared by the objects created on that device.
class Device {
public:
    explicit Device(int index = 0)
    {
        ws_context *c = nullptr;
        const int rc = ws_create(index, &c);
        if (rc != WS_OK) throw Error(rc, ws_last_error(nullptr));
        ctx_.reset(c, ws_destroy);
    }
    ws_context *get() const { return ctx_.get(); }
    static Device &shared()
    {
        static Device d(0);
        return d;
    }

private:
    std::shared_ptr<ws_context> ctx_;
};

namespace detail {
inline ws_image to_c(const Image8UC3 &m)
{
    ws_image im;
    im.data = m.data;
    im.width = m.cols;
    im.height = m.rows;
    im.stride = static_cast<int>(m.step);
    return im;
}
inline MatF64 run(Device &dev, const ws_params &p, const Image8UC3 &l, const Image8UC3 &r)
{
    const bool left = p.view == WS_VIEW_LEFT;
    MatF64 out(left ? l.rows : r.rows, left ? l.cols : r.cols);
    const ws_image li = to_c(l), ri = to_c(r);
    const int rc = ws_search_host(dev.get(), &p, &li, &ri, out.ptr(), out.cols, WS_OUT_F64);
    if (rc != WS_OK) throw Error(rc, ws_last_error(dev.get()));
    return out;
}
} // namespace detail

// BlockSearch (BlockSearch.h:9-46).  `cost` and `subpixel` are the build's extensions.
class BlockSearch {
public:
    BlockSearch(const Image8UC3 &leftImage, const Image8UC3 &rightImage, int blockSize,
                int minDisparity, int maxDisparity, Device &device = Device::shared())
        : leftImage_(leftImage), rightImage_(rightImage), blockSize_(blockSize),
          maxDisparity_(maxDisparity), minDisparity_(minDisparity), device_(device)
    {
    }

    int cost = WS_COST_SSD; // the reference's NORM_L2
    bool subpixel = false;

    MatF64 computeDisparityMapLeft(double smoothFactor) // BlockSearch.cpp:24-86
    {
        ws_params p = params(WS_VIEW_LEFT, smoothFactor);
        return detail::run(device_, p, leftImage_, rightImage_);
    }

    MatF64 computeDisparityMapRight(double smoothFactor, bool varBlock = false,
                                    double thres = 19.0) // BlockSearch.cpp:88-179
    {
        ws_params p = params(WS_VIEW_RIGHT, smoothFactor);
        p.var_block = varBlock;
        p.thres = thres;
        return detail::run(device_, p, leftImage_, rightImage_);
    }

private:
    ws_params params(int view, double smoothFactor) const
    {
        ws_params p;
        ws_params_default(&p);
        p.view = view;
        p.cost = cost;
        p.block_size = blockSize_;
        p.min_disparity = minDisparity_;
        p.max_disparity = maxDisparity_;
        p.smooth_factor = smoothFactor;
        p.subpixel = subpixel;
        return p;
    }
    Image8UC3 leftImage_, rightImage_;
    int blockSize_, maxDisparity_, minDisparity_;
    Device &device_;
};

// LinearSearch (LinearSearch.h:9-20).
class LinearSearch {
public:
    LinearSearch(const Image8UC3 &leftImage, const Image8UC3 &rightImage,
                 Device &device = Device::shared())
        : leftImage(leftImage), rightImage(rightImage), device_(device)
    {
    }
    MatF64 computeDisparityMap(double smoothFactor) // LinearSearch.cpp:10-59
    {
        ws_params p;
        ws_params_default(&p);
        p.view = WS_VIEW_LINEAR;
        p.smooth_factor = smoothFactor;
        return detail::run(device_, p, leftImage, rightImage);
    }

private:
    Image8UC3 leftImage, rightImage;
    Device &device_;
};

// The boundary methods of ImageRectifier (rectification.cpp:66-88, getters :515-521): block search
// on the rectified pair, then cv::warpPerspective(map, H_.inv(), original size, INTER_NEAREST) back
// to the original frame (the reference uses H_ for both views).  For pairs that are already
// rectified (Middlebury) leave H at the identity: the warp is then a copy and is skipped.
class RectifiedPair {
public:
    RectifiedPair(const Image8UC3 &leftRectified, const Image8UC3 &rightRectified,
                  Device &device = Device::shared())
        : left_(leftRectified), right_(rightRectified), device_(device)
    {
    }
    // H = the rectifying homography H_ (row-major 3x3); rows/cols = size of the original images
    void setHomography(const double H[9], int leftRows, int leftCols, int rightRows, int rightCols)
    {
        for (int i = 0; i < 9; ++i) H_[i] = H[i];
        rows_[0] = leftRows; cols_[0] = leftCols; rows_[1] = rightRows; cols_[1] = rightCols;
        warp_ = true;
    }
    void computeDisparityMapLeft(int blockSize, int minDisparity, int maxDisparity, double smoothFactor)
    {
        MatF64 rect = BlockSearch(left_, right_, blockSize, minDisparity, maxDisparity, device_)
                          .computeDisparityMapLeft(smoothFactor);
        disparityMapLeft = warp_ ? warpBack(rect, rows_[0], cols_[0]) : rect;
    }
    void computeDisparityMapRight(int blockSize, int minDisparity, int maxDisparity, double smoothFactor,
                                  bool varBlock = false, double thres = 10.0) // default thres: rectification.hpp:66
    {
        MatF64 rect = BlockSearch(left_, right_, blockSize, minDisparity, maxDisparity, device_)
                          .computeDisparityMapRight(smoothFactor, varBlock, thres);
        disparityMapRight = warp_ ? warpBack(rect, rows_[1], cols_[1]) : rect;
    }
    const MatF64 &getDisparityMapLeft() const { return disparityMapLeft; }
    const MatF64 &getDisparityMapRight() const { return disparityMapRight; }

private:
    MatF64 warpBack(const MatF64 &rect, int rows, int cols)
    {
        // H_.inv() by adjugate / determinant, handed on as the warpPerspective matrix
        const double *m = H_;
        const double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
                         m[2] * (m[3] * m[7] - m[4] * m[6]);
        const double r = 1.0 / d;
        const double inv[9] = {(m[4] * m[8] - m[5] * m[7]) * r, (m[2] * m[7] - m[1] * m[8]) * r, (m[1] * m[5] - m[2] * m[4]) * r,
                               (m[5] * m[6] - m[3] * m[8]) * r, (m[0] * m[8] - m[2] * m[6]) * r, (m[2] * m[3] - m[0] * m[5]) * r,
                               (m[3] * m[7] - m[4] * m[6]) * r, (m[1] * m[6] - m[0] * m[7]) * r, (m[0] * m[4] - m[1] * m[3]) * r};
        MatF64 out(rows, cols);
        const int rc = ws_warp_nearest_host(device_.get(), rect.ptr(), rect.cols, rect.rows, rect.cols, inv,
                                            out.ptr(), cols, rows, cols);
        if (rc != WS_OK) throw Error(rc, ws_last_error(device_.get()));
        return out;
    }
    Image8UC3 left_, right_;
    Device &device_;
    double H_[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    int rows_[2] = {0, 0}, cols_[2] = {0, 0};
    bool warp_ = false;
    MatF64 disparityMapLeft, disparityMapRight;
};

// A CV_32FC1 map owned by value (what main.cpp:50-64 hands from stage to stage).
class MatF32 {
public:
    MatF32() = default;
    MatF32(int rows, int cols) : rows(rows), cols(cols), buf_(static_cast<size_t>(rows) * cols, 0.0f) {}
    int rows = 0, cols = 0;
    float &at(int y, int x) { return buf_[static_cast<size_t>(y) * cols + x]; }
    float at(int y, int x) const { return buf_[static_cast<size_t>(y) * cols + x]; }
    float *ptr() { return buf_.data(); }
    const float *ptr() const { return buf_.data(); }

private:
    std::vector<float> buf_;
};

// Reconstruction/reconstruction.h:26-33 with the same names and argument order.
inline void removeDisparityOutliers(MatF32 &disparityMap, int kernelSize, float thrFront, float thrBack,
                                    Device &device = Device::shared())
{
    const int rc = ws_remove_disparity_outliers(device.get(), disparityMap.ptr(), disparityMap.cols, disparityMap.rows,
                                                disparityMap.cols, kernelSize, thrFront, thrBack);
    if (rc != WS_OK) throw Error(rc, ws_last_error(device.get()));
}

inline MatF32 convertDisparityToDepth(const MatF32 &dispImage, float focalLength, float baseline,
                                      Device &device = Device::shared())
{
    MatF32 depth(dispImage.rows, dispImage.cols);
    const int rc = ws_convert_disparity_to_depth(device.get(), dispImage.ptr(), dispImage.cols, dispImage.rows,
                                                 dispImage.cols, focalLength, baseline, depth.ptr(), depth.cols);
    if (rc != WS_OK) throw Error(rc, ws_last_error(device.get()));
    return depth;
}

// reconstruction(bgrImage, depthValues, intrinsics, thrMesh) (reconstruction.cpp:152-208); the
// reference's hard-coded output path becomes an argument.
inline void reconstruction(const Image8UC3 &bgrImage, const MatF32 &depthValues, const float intrinsics[9],
                           float thrMesh, const std::string &meshPath, Device &device = Device::shared())
{
    const size_t n = static_cast<size_t>(depthValues.rows) * depthValues.cols;
    std::vector<float> positions(4 * n);
    std::vector<uint8_t> colors(4 * n);
    const ws_image img = detail::to_c(bgrImage);
    int rc = ws_back_project(device.get(), depthValues.ptr(), depthValues.cols, depthValues.rows, depthValues.cols,
                             intrinsics, &img, positions.data(), colors.data());
    if (rc != WS_OK) throw Error(rc, ws_last_error(device.get()));
    rc = ws_write_mesh_off(meshPath.c_str(), positions.data(), colors.data(), depthValues.cols, depthValues.rows, thrMesh);
    if (rc != WS_OK) throw Error(rc, "Failed to write mesh! Check file path!");
}

#ifdef WSAMD_WITH_OPENCV
inline Image8UC3 view(const cv::Mat &m)
{
    CV_Assert(m.type() == CV_8UC3);
    return view(m.data, m.rows, m.cols, m.step);
}
inline cv::Mat to_cv(const MatF64 &m)
{
    cv::Mat out(m.rows, m.cols, CV_64F);
    for (int y = 0; y < m.rows; ++y)
        for (int x = 0; x < m.cols; ++x) out.at<double>(y, x) = m.at(y, x);
    return out;
}
#endif

} // namespace wsamd
