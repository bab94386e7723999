// Compile-and-run check of the WSAMD_WITH_OPENCV adapters against tests/cxx/opencv_stub (NOT OpenCV; see that header).
// No device needed: only the header-only adapters run.
#define WSAMD_WITH_OPENCV
#include "stereo_reconstruction_amd/host/window_search.hpp"

#include <cstdio>

int main()
{
    cv::Mat img(4, 5, CV_8UC3);
    for (int i = 0; i < 4 * 5 * 3; ++i) img.data[i] = (uint8_t)i;
    const wsamd::Image8UC3 v = wsamd::view(img);
    if (v.data != img.data || v.rows != 4 || v.cols != 5 || v.step != 15) return 1;
    wsamd::MatF64 m(2, 3);
    const double vals[6] = {1.0, 2.0, 3.0, -4.0, 5.5, 6.0};
    for (int i = 0; i < 6; ++i) m.at(i / 3, i % 3) = vals[i];
    const cv::Mat out = wsamd::to_cv(m);
    if (out.type() != CV_64F || out.rows != 2 || out.cols != 3 || out.at<double>(1, 0) != -4.0 || out.at<double>(1, 1) != 5.5) return 2;
    bool threw = false;
    try { (void)wsamd::view(out); } catch (const std::exception &) { threw = true; } // not CV_8UC3
    if (!threw) return 3;
    std::puts("opencv adapters ok (against the stub)");
    return 0;
}
