// Test driver for the C++ facade (stereo_reconstruction_amd/host/window_search.hpp): reads two raw
// BGR images, runs the reference-named classes, writes the CV_64F-like map as raw doubles.
// usage: facade_driver left.raw w1 h1 right.raw w2 h2 view(left|right|linear) bs minD maxD cost(ssd|sad) out.raw
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "stereo_reconstruction_amd/host/window_search.hpp"

static std::vector<uint8_t> slurp(const char *path, size_t n)
{
    std::vector<uint8_t> v(n);
    FILE *f = fopen(path, "rb");
    if (!f || fread(v.data(), 1, n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc != 13) { fprintf(stderr, "bad usage\n"); return 2; }
    const int w1 = atoi(argv[2]), h1 = atoi(argv[3]), w2 = atoi(argv[5]), h2 = atoi(argv[6]);
    const std::string mode = argv[7];
    const int bs = atoi(argv[8]), minD = atoi(argv[9]), maxD = atoi(argv[10]);
    std::vector<uint8_t> l = slurp(argv[1], (size_t)w1 * h1 * 3), r = slurp(argv[4], (size_t)w2 * h2 * 3);
    try {
        wsamd::Image8UC3 left = wsamd::view(l.data(), h1, w1), right = wsamd::view(r.data(), h2, w2);
        wsamd::MatF64 out;
        if (mode == "linear") {
            out = wsamd::LinearSearch(left, right).computeDisparityMap(1.0);
        } else if (mode == "rectifier") { // the ImageRectifier boundary, both maps; writes the left one
            wsamd::RectifiedPair pair(left, right);
            pair.computeDisparityMapLeft(bs, minD, maxD, 1.0);
            pair.computeDisparityMapRight(bs, minD, maxD, 1.0);
            out = pair.getDisparityMapLeft();
        } else {
            wsamd::BlockSearch search(left, right, bs, minD, maxD);
            search.cost = strcmp(argv[11], "sad") == 0 ? WS_COST_SAD : WS_COST_SSD;
            out = mode == "left" ? search.computeDisparityMapLeft(1.0) : search.computeDisparityMapRight(1.0);
        }
        FILE *f = fopen(argv[12], "wb");
        if (!f || fwrite(out.ptr(), sizeof(double), (size_t)out.rows * out.cols, f) != (size_t)out.rows * out.cols) return 3;
        fclose(f);
        printf("%d %d\n", out.cols, out.rows);
    } catch (const wsamd::Error &e) {
        fprintf(stderr, "wsamd::Error %d: %s\n", e.code(), e.what());
        return 10 - e.code();
    }
    return 0;
}
