// Compile check of ms_shim.hpp against a GpuMat-shaped struct (OpenCV is not in this image).
#include "ms_shim.hpp"
struct FakeGpuMat {                       // field names/types of cv::cuda::GpuMat (core/cuda.hpp:283-303)
    int flags = 0, rows = 0, cols = 0;
    size_t step = 0;
    unsigned char *data = nullptr;
    int type() const { return flags; }
    void create(int r, int c, int t) { rows = r; cols = c; flags = t; }      // GpuMat::create(rows, cols, type)
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
        const ms_rig rig = msshim::calibrateCameras(6, 1920, 1080);          // calibrateCameras + the scales of stitch_calib
        msshim::Calibration cal;
        std::unique_ptr<msshim::Compositor> calibrated = msshim::stitch_calib(frames, MS_PROJ_CYLINDRICAL, true, cal);      // stitch_calib
        (void)rig; (void)calibrated;
        comp.stitch_one(frames, &a, (FakeGpuMat *)nullptr);
        for (int i = 0; i < 6; ++i) comp.feed_online(frames[i], i);
        comp.blend(&a, (FakeGpuMat *)nullptr);
        std::vector<FakeGpuMat> slabs(1);
        comp.stitch_one_i420(frames, slabs);
        // MeshWarper over ImageFeatures / MatchesInfo shaped types
        struct Pt { float x, y; }; struct KeyPoint { Pt pt; }; struct Size { int width, height; };
        struct Features { Size img_size; std::vector<KeyPoint> keypoints; };
        struct DMatch { int queryIdx, trainIdx; };
        struct Matches { int src_img_idx, dst_img_idx; std::vector<DMatch> matches; std::vector<unsigned char> inliers_mask; int num_inliers; };
        std::vector<Features> feats(6);
        std::vector<Matches> pairwise(6);
        msshim::knnRatioMatches(a, b, [&](int q, int t, float d) { pairwise[0].matches.push_back({q, t}); (void)d; });
        // featurefinder::findFeatures / matchFeatures (featurefinder.h:7-13) on the device front-end, feeding the mesh warper
        std::vector<FakeGpuMat> fmasks;
        msshim::featurefinder::featureMasks(frames, fmasks);
        std::vector<msshim::featurefinder::ImageFeatures<FakeGpuMat>> ffeats, fprev;
        msshim::featurefinder::findFeatures(frames, fmasks, ffeats);
        std::vector<msshim::featurefinder::MatchesInfo> fpair(6);
        msshim::featurefinder::matchFeatures(ffeats, fpair);
        msshim::featurefinder::matchFeaturesTemporal(ffeats, ffeats, fpair);
        msshim::MeshWarper mw(6, 10, 10, 600.f, 1.0, 0.5);
        mw.calibrateMeshWarp(comp, frames, ffeats, fpair);
        mw.calibrateMeshWarp(comp, frames, feats, pairwise);
    } catch (const msshim::Error &e) {
        return e.code;
    }
    return 0;
}
