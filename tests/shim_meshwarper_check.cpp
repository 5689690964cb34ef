// Host-only check of msshim::MeshWarper's match filtering / selection state machine (360_stitcher/meshwarper.cpp:158-335, :888-946)
// against hand-derived expectations.  No device work: select() never reaches ms_create_mesh.  Built and run by tests/test_abi.py.
#include <cstdio>
#include "../video-stitcher_amd/shim/ms_shim.hpp"

struct Pt { float x, y; };
struct KeyPoint { Pt pt; };
struct Size { int width, height; };
struct Features { Size img_size; std::vector<KeyPoint> keypoints; };                 // cv::detail::ImageFeatures
struct DMatch { int queryIdx, trainIdx; };
struct Matches { int src_img_idx, dst_img_idx; std::vector<DMatch> matches; std::vector<unsigned char> inliers_mask; int num_inliers; };   // MatchesInfo

#define EXPECT(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static Matches pair_of(std::vector<Features> &f, int src, int dst, const std::vector<float> &xy)   // xy: x1 y1 x2 y2 inlier, ...
{
    Matches m{src, dst, {}, {}, 0};
    for (size_t k = 0; k + 4 < xy.size() + 0 && k + 5 <= xy.size(); k += 5) {
        f[src].keypoints.push_back({{xy[k], xy[k + 1]}});
        f[dst].keypoints.push_back({{xy[k + 2], xy[k + 3]}});
        m.matches.push_back({(int)f[src].keypoints.size() - 1, (int)f[dst].keypoints.size() - 1});
        m.inliers_mask.push_back(xy[k + 4] != 0);
        m.num_inliers += xy[k + 4] != 0;
    }
    return m;
}

int main()
{
    const int n = 6;
    const float f = 600.f;                      // theta * f: (src, dst) = (1, 0): -1 * 600
    msshim::MeshWarper w(n, 10, 10, f, 1.0, 1.0);
    EXPECT(w.theta(1, 0) == -1.f && w.theta(0, 5) == -1.f && w.theta(3, 2) == 4.25f && w.theta(4, 3) == -0.25f);

    std::vector<Features> feat(n, Features{{960, 627}, {}});
    std::vector<Matches> pw;
    // kept pair (src = dst + 1): expected x offset p1.x - p2.x = theta * f = -600
    pw.push_back(pair_of(feat, 1, 0, {100, 50, 700, 60, 1,       // ok
                                       100, 50, 700, 95, 1,       // |dy| = 45 > 40: dropped
                                       100, 50, 1005, 50, 1,      // | -600 - (-905) | = 305 > 300: dropped
                                       120, 80, 715, 75, 0,       // not an inlier: dropped
                                       130, 90, 729, 91, 1}));    // ok
    // a pair with src != dst + 1 is ignored, except (0 -> n - 1), the wrap-around seam
    pw.push_back(pair_of(feat, 0, 1, {100, 50, 700, 60, 1}));
    pw.push_back(pair_of(feat, 0, 5, {50, 40, 652, 44, 1}));      // theta = -1: offset -602, ok
    std::vector<std::vector<ms_mesh_match>> filt;
    w.filterMatches(pw, feat, filt);
    EXPECT(filt[1].size() == 2 && filt[1][0].x2 == 700 && filt[1][1].y2 == 91 && filt[1][0].dst == 0);
    EXPECT(filt[0].size() == 1 && filt[0][0].dst == 5);
    for (int v = 2; v < n; ++v) EXPECT(filt[v].empty());

    // first calibration: no history.  View 1's seam moved |100+130|/2 - ... from prev_avg = 0 by more than RECALIB_THRESH -> new matches used
    auto use = w.select(feat, pw);
    EXPECT(use[1].size() == 2 && use[0].size() == 1);
    // second calibration, same features: the seam did not move -> the stored (old) matches are used, and they are the same lists
    auto use2 = w.select(feat, pw);
    EXPECT(use2[1].size() == 2 && use2[1][1].x1 == 130 && use2[0].size() == 1);
    // third calibration: view 1's matches shift by 10 px (< RECALIB_THRESH): old matches are still used (x1 stays 100 / 130)
    std::vector<Features> feat3(n, Features{{960, 627}, {}});
    std::vector<Matches> pw3;
    pw3.push_back(pair_of(feat3, 1, 0, {110, 50, 700, 60, 1, 140, 90, 729, 91, 1}));
    pw3.push_back(pair_of(feat3, 0, 5, {50, 40, 652, 44, 1}));
    auto use3 = w.select(feat3, pw3);
    EXPECT(use3[1].size() == 2 && use3[1][0].x1 == 100 && use3[1][1].x1 == 130);
    // fourth: a 60 px shift (> RECALIB_THRESH): the new matches are used and become the stored ones
    std::vector<Features> feat4(n, Features{{960, 627}, {}});
    std::vector<Matches> pw4;
    pw4.push_back(pair_of(feat4, 1, 0, {160, 50, 700, 60, 1, 190, 90, 729, 91, 1}));
    pw4.push_back(pair_of(feat4, 0, 5, {50, 40, 652, 44, 1}));
    auto use4 = w.select(feat4, pw4);
    EXPECT(use4[1].size() == 2 && use4[1][0].x1 == 160 && use4[1][1].x1 == 190);
    auto use5 = w.select(feat4, pw4);
    EXPECT(use5[1][0].x1 == 160);
    // the per-image cap
    std::vector<Features> featc(n, Features{{960, 627}, {}});
    std::vector<float> many;
    for (int k = 0; k < 150; ++k) { const float v[5] = {100.f + k, 50, 700.f + k, 52, 1}; many.insert(many.end(), v, v + 5); }
    std::vector<Matches> pwc{pair_of(featc, 2, 1, many)};
    msshim::MeshWarper w2(n, 10, 10, f, 1.0, 1.0);
    EXPECT(w2.select(featc, pwc)[2].size() == (size_t)msshim::MeshWarper::MAX_FEATURES_PER_IMAGE);
    std::printf("ok\n");
    return 0;
}
