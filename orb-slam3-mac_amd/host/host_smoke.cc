// host_smoke.cc -- exercises the signature-preserving C++ classes end to end (needs a GPU to run;
// compiles anywhere).  Built and run by tests/test_gpu_host_cpp.py.
#include <cstdio>
#include <cstdlib>
#include "ORBextractor.h"
#include "ORBmatcher.h"

extern "C" void synth_frame(uint8_t *out, int w, int h, int stride, unsigned long long seed, int frame_id);

int main()
{
    const int W = 640, H = 480;
    cv::Mat im(H, W, cv::CV_8U), mask, desc;
    synth_frame(im.data, W, H, W, 7ull, 0);
    ORB_SLAM3::ORBextractor ex(1000, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> kps;
    std::vector<int> lap = {0, 1000};
    int mono = ex(im, mask, kps, desc, lap);                       // Frame.cc:302 / :412-416 call shape
    if (mono != 0 || kps.size() < 800 || desc.rows != (int)kps.size()) { printf("FAIL extract %d %zu\n", mono, kps.size()); return 1; }
    cv::Mat empty;
    if (ex(empty, mask, kps, desc, lap) != -1) { printf("FAIL empty\n"); return 1; }
    ex(im, mask, kps, desc, lap);
    ex.SyncImagePyramid();
    if (ex.mvImagePyramid.size() != 8 || ex.mvImagePyramid[0].cols != W || ex.mvImagePyramid[7].cols != 179) { printf("FAIL pyramid\n"); return 1; }
    if (ex.mvImagePyramid[0].ptr(5)[7] != im.ptr(5)[7]) { printf("FAIL pyramid content\n"); return 1; }
    if (ex.GetLevels() != 8 || ex.GetScaleFactors()[1] != 1.2f) { printf("FAIL getters\n"); return 1; }
    int d = ORB_SLAM3::ORBmatcher::DescriptorDistance(desc.row(0), desc.row(0));
    int d2 = ORB_SLAM3::ORBmatcher::DescriptorDistance(desc.row(0), desc.row(1));
    if (d != 0 || d2 <= 0) { printf("FAIL distance\n"); return 1; }
    printf("HOST_CPP_OK %zu keypoints, d01=%d\n", kps.size(), d2);
    return 0;
}
