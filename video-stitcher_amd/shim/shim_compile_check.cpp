// Compile check of ms_shim.hpp against a GpuMat-shaped struct (OpenCV is not in this image).
#include "ms_shim.hpp"
struct FakeGpuMat {                       // field names/types of cv::cuda::GpuMat (core/cuda.hpp:283-303)
    int flags = 0, rows = 0, cols = 0;
    size_t step = 0;
    unsigned char *data = nullptr;
    int type() const { return flags; }
};
int shim_compile_check()
{
    FakeGpuMat a, b, c, d;
    try {
        msshim::cuda::remap(a, b, c, d, MS_INTER_LINEAR);
        msshim::cuda::pyrDown(a, b);
        msshim::cuda::convertTo(a, b, 1.02);
        msshim::addSrcWeightGpu32F(a, b, c, d, 1, 1);
        msshim::Compositor comp(6, 1920, 1080, MS_PROJ_CYLINDRICAL, 611.f, 5, false, 3840, 1920);
        std::vector<FakeGpuMat> frames(6);
        comp.stitch_one(frames, &a, (FakeGpuMat *)nullptr);
        for (int i = 0; i < 6; ++i) comp.feed_online(frames[i], i);
        comp.blend(&a, (FakeGpuMat *)nullptr);
    } catch (const msshim::Error &e) {
        return e.code;
    }
    return 0;
}
