// pipeline_main.cpp -- the reference's src/main.cpp:13-66 on an already rectified pair, with its
// stages on the MI355X: load -> (pose estimation + rectification skipped: Middlebury pairs are
// rectified) -> computeDisparityMapRight(17, 0, 200, 0.9) -> hand-off (PFM instead of the
// saturating 8-bit PNG, same 0..255 clamp as the PNG round trip) -> removeDisparityOutliers(500,
// 1.5, 0.8) -> convertDisparityToDepth(f, 1) -> reconstruction(...) -> OFF mesh.
//
// usage: pipeline_main im0.ppm im1.ppm calib.txt out_prefix
// build: g++ -std=c++17 -O2 -I<repo> examples/pipeline_main.cpp -L<repo>/stereo_reconstruction_amd -lws_stereo
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <string>

#include "stereo_reconstruction_amd/host/window_search.hpp"

int main(int argc, char **argv)
{
    if (argc != 5) {
        fprintf(stderr, "usage: %s left.ppm right.ppm calib.txt out_prefix\n", argv[0]);
        return 2;
    }
    uint8_t *l = nullptr, *r = nullptr;
    int w1, h1, w2, h2;
    ws_calib calib;
    if (ws_ppm_read(argv[1], &l, &w1, &h1) != WS_OK || ws_ppm_read(argv[2], &r, &w2, &h2) != WS_OK ||
        ws_calib_read(argv[3], &calib) != WS_OK) {
        fprintf(stderr, "cannot read the inputs\n");
        return 1;
    }
    const std::string prefix = argv[4];
    try {
        const wsamd::Image8UC3 left = wsamd::view(l, h1, w1), right = wsamd::view(r, h2, w2);

        // 2. disparity (main.cpp:38-42)
        wsamd::RectifiedPair rectifier(left, right);
        rectifier.computeDisparityMapRight(17, 0, 200, 0.9);
        const wsamd::MatF64 &disp = rectifier.getDisparityMapRight();

        // hand-off: what imwrite(8-bit) + readGrayscaleImageAsDisparityMap keep of it (main.cpp:42,50)
        wsamd::MatF32 disparityImage(disp.rows, disp.cols);
        for (int y = 0; y < disp.rows; ++y)
            for (int x = 0; x < disp.cols; ++x)
                disparityImage.at(y, x) = (float)std::min(255.0, std::max(0.0, std::nearbyint(disp.at(y, x))));
        if (ws_pfm_write((prefix + "_disparity.pfm").c_str(), disparityImage.ptr(), disp.cols, disp.rows, disp.cols) != WS_OK)
            return 1;

        // 3. reconstruct (main.cpp:53-64)
        wsamd::removeDisparityOutliers(disparityImage, 500, 1.5f, 0.8f);
        const float focalLength = calib.cam1[0];
        wsamd::MatF32 depthValues = wsamd::convertDisparityToDepth(disparityImage, focalLength, 1.0f);
        wsamd::reconstruction(right, depthValues, calib.cam1, 1.0f, prefix + "_mesh.off");
        printf("%d %d\n", disp.cols, disp.rows);
    } catch (const wsamd::Error &e) {
        fprintf(stderr, "wsamd::Error %d: %s\n", e.code(), e.what());
        return 1;
    }
    ws_free(l);
    ws_free(r);
    return 0;
}
