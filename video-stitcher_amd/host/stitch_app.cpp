// stitch_app.cpp -- C++ host pipeline over the C-ABI, with the thread / queue graph of the reference's main loop
// (360_stitcher/timed.cpp): capture -> LockableVector of the newest frame per camera -> stitch_one (upload, compositor,
// push) -> BlockingQueue -> consume (8U panorama, optional I420 for the encoder) and a recalibration thread that swaps
// CPW meshes while frames keep flowing.  No OpenCV: device images are plain HIP allocations wrapped as ms_image.
//
//   reference                                   here
//   ------------------------------------------  ---------------------------------------------------------------
//   stitch_calib / warpImages (calibration.cpp)  calibrate(): ms_set_camera/gain, ms_build_maps, ms_build_masks, ms_init_blender
//   LockableVector<Mat> imgs (lockablevector.h)  Lockable<std::vector<HostFrame>>
//   capture threads (networking.cpp / debug)     capture(): synthetic frames (same pattern as video-stitcher_amd/synth.py, noise off);
//                                                with --nv12 the cameras deliver NV12 (defs.h:10-17) and cvtColor(YUV2BGR_NV12)
//                                                (networking.cpp:45-47, CPU in the reference) runs on the device after a half-size upload
//   stitch_one (timed.cpp:123-152)               stitch_one(): hipMemcpy2DAsync x N + msshim::Compositor::stitch_one + results.push
//   BlockingQueue<GpuMat> results                BlockingQueue<Slot*>
//   consume (timed.cpp:232-330)                  consume(): optional ms_bgr_to_i420, download, checksum / dump
//   recalibrate thread (timed.cpp:414-463)       recalibrate(): ms_set_mesh per view from its own stream
//
// Usage: stitch_app [--views 6] [--size 1920x1080] [--out 3840x1920] [--hfov 90] [--bands 5] [--frames 300] [--cpw]
//                   [--i420] [--nv12 | --nv12-direct] [--dump pano.bin] [--no-upload] [--solve-mesh]
//                   [--reference-calib [--work-megapix 0.6] [--seam-megapix 0.01] [--compose-megapix 1.4]]
// --reference-calib runs msshim::stitch_calib (calibration.cpp:252-311): the reference's rig and scale bookkeeping, cylindrical warper, seam-scale
// gains + Voronoi seams from the first frames, the num_bands rule, and -- with the default COMPOSE_MEGAPIX -- cuda::resize of every frame.
// Prints one JSON line: end-to-end frames/s INCLUDING the PCIe upload of every source frame (unlike bench.py).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../shim/ms_shim.hpp"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// ---- the two containers of the reference's thread graph -------------------------------------------------------
template <class T> struct Lockable {           // lockablevector.h: a value + the mutex every user takes
    T v;
    std::mutex mu;
};
template <class T> class BlockingQueue {       // blockingqueue.h: unbounded push, blocking pop
public:
    void push(T x) { { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(x)); } cv_.notify_one(); }
    T pop() { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return !q_.empty(); }); T x = std::move(q_.front()); q_.pop_front(); return x; }
private:
    std::mutex mu_; std::condition_variable cv_; std::deque<T> q_;
};

// ---- a device image with the fields the shim expects (cv::cuda::GpuMat's) -------------------------------------
struct DevMat {
    int rows = 0, cols = 0, flags = 0;
    size_t step = 0;
    unsigned char *data = nullptr;
    int type() const { return flags; }
    void create(int r, int c, int t) { create(r, c, t, t == MS_8UC3 ? 3 : (t == MS_32FC1 ? 4 : (t == MS_16SC3 ? 6 : (t == MS_16SC1 ? 2 : 1)))); }    // GpuMat::create(rows, cols, type)
    void create(int r, int c, int t, int elem, bool contiguous = false)
    {
        if (data && r == rows && c == cols && t == flags) return;       // GpuMat::create is a no-op when nothing changes
        if (data) HIPCHECK(hipFree(data));
        size_t pitch = (size_t)c * elem;
        if (contiguous) HIPCHECK(hipMalloc((void **)&data, pitch * r));  // (the I420 planes are one contiguous buffer, like the Mat cvtColor fills)
        else HIPCHECK(hipMallocPitch((void **)&data, &pitch, (size_t)c * elem, r));
        rows = r; cols = c; flags = t; step = pitch;
    }
};
struct HostFrame { unsigned char *p = nullptr; int w = 0, h = 0; long long seq = -1; };   // pinned BGR frame

struct Options {
    int views = 6, w = 1920, h = 1080, out_w = 3840, out_h = 1920, bands = 5, frames = 300;
    double hfov = 90.0;
    bool cpw = false, i420 = false, upload = true, nv12 = false, solve_mesh = false;
    bool nv12_direct = false;         // --nv12-direct: no cvtColor pass at all, the warp samples the NV12 planes (ms_stitch_nv12)
    int consume_w = 0, consume_h = 0;           // > 0: consume()'s resize + black bars + BGR2YUV_I420 on the device (timed.cpp:251-316), OUTPUT_WIDTH x OUTPUT_HEIGHT
    int update_mask = 0;                        // > 0: mb->update_mask(idx, ...) after every mesh swap (timed.cpp:598-605, commented out there), enqueue-only with this margin
    bool reference_calib = false;               // stitch_calib as the reference ships it: cylindrical warper, megapixel budgets of defs.h, seam-scale pipeline
    double work_mp = 0.6, seam_mp = 0.01, compose_mp = 1.4;      // WORK_MEGAPIX, SEAM_MEAGPIX, COMPOSE_MEGAPIX (defs.h:51-53)
    std::string dump;
};

// same pattern as synth.frame(w, h, i, t, noise=False): clip(rint(128 + 60 sin(2pi(x/97 + y/61 + i/7 + c/3)) + 40 checker))
static void synth_frame(unsigned char *dst, int w, int h, int view)
{
    const double two_pi = 2.0 * M_PI;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const double phase = x / 97.0 + y / 61.0 + view / 7.0;
            const double chk = 40.0 * (((x / 32) + (y / 32)) & 1);
            for (int c = 0; c < 3; ++c) {
                double v = 128.0 + 60.0 * std::sin(two_pi * (phase + c / 3.0)) + chk;
                v = std::nearbyint(v);
                dst[((size_t)y * w + x) * 3 + c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
}

// A scene-consistent synthetic camera frame for --solve-mesh: a texture fixed on the viewing sphere (blocks of random grey levels: corners for FAST,
// structure for the descriptors) seen through the pinhole camera `view` of the rig, so that neighbouring warped views show the SAME content where they
// overlap and the device front-end finds real correspondences.  `shift_px` moves this camera's view of the scene by that many panorama pixels
// (a stand-in for parallax): it is what the mesh optimiser then has to absorb.
static void synth_scene_frame(unsigned char *dst, int w, int h, int view, int n_views, double hfov_deg, double pano_scale, double shift_px)
{
    const float rot = (float)(2.0 * M_PI * (double)(float)view / n_views);
    const double c = std::cos((double)rot), s = std::sin((double)rot);
    const double f = (w / 2.0) / std::tan(hfov_deg * M_PI / 180.0 / 2.0);
    const double cell = 32.0 / pano_scale;                          // blocks of 32 panorama pixels
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const double dx = (x - w / 2.0) / f, dy = (y - h / 2.0) / f, dz = 1.0;
            const double X = c * dx + s * dz, Y = dy, Z = -s * dx + c * dz;           // world = R_y(rot) * camera
            const double u = std::atan2(X, Z) + shift_px / pano_scale, v = M_PI - std::acos(Y / std::sqrt(X * X + Y * Y + Z * Z));
            const long long cu = (long long)std::floor(u / cell), cv = (long long)std::floor(v / cell);
            const double fu = u / cell - cu, fv = v / cell - cv;
            for (int ch = 0; ch < 3; ++ch) {
                unsigned hsh = (unsigned)(cu * 73856093ll) ^ (unsigned)(cv * 19349663ll) ^ (unsigned)(ch * 83492791);
                hsh ^= hsh >> 13; hsh *= 0x5bd1e995u; hsh ^= hsh >> 15;
                double val = 40.0 + (double)((hsh >> 8) & 0xff) * 175.0 / 255.0;
                val += 18.0 * std::sin(2.0 * M_PI * (fu + 0.3 * ch)) * std::sin(2.0 * M_PI * fv);      // some texture inside a block
                val = std::nearbyint(val);
                dst[((size_t)y * w + x) * 3 + ch] = (unsigned char)(val < 1 ? 1 : (val > 255 ? 255 : val));    // never 0: black marks "outside the view" in createMesh's masks
            }
        }
}

// synthetic NV12 camera frame, same pattern as synth.nv12_frame: Y = the green channel of synth_frame's pattern, interleaved U/V = smooth ramps
static void synth_nv12(unsigned char *dst, int w, int h, int view)
{
    const double two_pi = 2.0 * M_PI;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const double phase = x / 97.0 + y / 61.0 + view / 7.0;
            double v = 128.0 + 60.0 * std::sin(two_pi * (phase + 1.0 / 3.0)) + 40.0 * (((x / 32) + (y / 32)) & 1);
            v = std::nearbyint(v);
            dst[(size_t)y * w + x] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    unsigned char *uv = dst + (size_t)w * h;
    for (int y = 0; y < h / 2; ++y)
        for (int x = 0; x < w / 2; ++x) {
            uv[(size_t)y * w + 2 * x] = (unsigned char)(64 + (3 * x + 5 * view) % 128);
            uv[(size_t)y * w + 2 * x + 1] = (unsigned char)(64 + (2 * y + 7 * view) % 128);
        }
}

// a smooth synthetic CPW mesh (the optimiser that produces real ones is out of scope): identity + amp sin(2 pi u + phase) sin(pi v)
static void make_mesh(int aw, int ah, int N, int M, double phase, double amp, std::vector<float> &mx, std::vector<float> &my)
{
    mx.resize((size_t)N * M); my.resize((size_t)N * M);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < M; ++j) {
            const double u = (double)j / (M - 1), v = (double)i / (N - 1);
            const double d = amp * std::sin(2.0 * M_PI * u + phase) * std::sin(M_PI * v);
            mx[(size_t)i * M + j] = (float)(u * (aw - 1) + d);
            my[(size_t)i * M + j] = (float)(v * (ah - 1) + 0.5 * d);
        }
}

struct Slot {                 // one entry of the ring of caller-owned outputs (no per-frame allocation)
    DevMat pano8u, i420;
    hipEvent_t done = nullptr;
    long long seq = -1;
};

int main(int argc, char **argv)
{
    Options o;
    for (int a = 1; a < argc; ++a) {
        std::string k = argv[a];
        auto next = [&]() -> const char * { if (a + 1 >= argc) { fprintf(stderr, "missing value for %s\n", k.c_str()); exit(2); } return argv[++a]; };
        if (k == "--views") o.views = atoi(next());
        else if (k == "--size") sscanf(next(), "%dx%d", &o.w, &o.h);
        else if (k == "--out") sscanf(next(), "%dx%d", &o.out_w, &o.out_h);
        else if (k == "--hfov") o.hfov = atof(next());
        else if (k == "--bands") o.bands = atoi(next());
        else if (k == "--frames") o.frames = atoi(next());
        else if (k == "--cpw") o.cpw = true;
        else if (k == "--solve-mesh") o.cpw = o.solve_mesh = true;
        else if (k == "--update-mask") { o.cpw = true; o.update_mask = atoi(next()); }
        else if (k == "--consume") sscanf(next(), "%dx%d", &o.consume_w, &o.consume_h);
        else if (k == "--i420") o.i420 = true;
        else if (k == "--no-upload") o.upload = false;
        else if (k == "--nv12") o.nv12 = true;
        else if (k == "--nv12-direct") { o.nv12 = true; o.nv12_direct = true; }
        else if (k == "--dump") o.dump = next();
        else if (k == "--reference-calib") o.reference_calib = true;
        else if (k == "--work-megapix") o.work_mp = atof(next());
        else if (k == "--seam-megapix") o.seam_mp = atof(next());
        else if (k == "--compose-megapix") o.compose_mp = atof(next());
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fprintf(stderr, "stitch_app: no HIP device (libmsstitch has no CPU fallback)\n"); return 3; }

    try {
        // ---- stitch_calib ------------------------------------------------------------------------------------
        std::unique_ptr<msshim::Compositor> comp_owner;
        msshim::Calibration cal;
        if (o.reference_calib) {
            // the reference's own calibration from the first frame of every camera (device 8UC3, full size)
            std::vector<DevMat> first(o.views);
            std::vector<unsigned char> host((size_t)o.w * o.h * 3);
            for (int i = 0; i < o.views; ++i) {
                first[i].create(o.h, o.w, MS_8UC3, 3);
                synth_frame(host.data(), o.w, o.h, i);
                HIPCHECK(hipMemcpy2D(first[i].data, first[i].step, host.data(), (size_t)o.w * 3, (size_t)o.w * 3, o.h, hipMemcpyHostToDevice));
            }
            comp_owner = msshim::stitch_calib(first, MS_PROJ_CYLINDRICAL, o.cpw, cal, o.hfov, o.work_mp, o.seam_mp, o.compose_mp, 5.f, -1, -1, 1, -1, nullptr, o.cpw ? o.update_mask : 0);
            for (auto &m : first) HIPCHECK(hipFree(m.data));
            const ms_pano_geom g = comp_owner->panoGeom();
            o.out_w = (2 * std::max(std::abs(cal.pano_roi.x), std::abs(cal.pano_roi.x + cal.pano_roi.width)) + 1) & ~1;
            o.out_h = (2 * std::max(std::abs(cal.pano_roi.y), std::abs(cal.pano_roi.y + cal.pano_roi.height)) + 1) & ~1;
            o.bands = g.num_bands;
            fprintf(stderr, "stitch_calib: work %.4f seam %.4f compose %.4f, frames %dx%d%s, warper scale %.3f, pano %dx%d, blend width %.1f -> %d bands, gains",
                    cal.rig.work_scale, cal.rig.seam_scale, cal.rig.compose_scale, cal.rig.compose_width, cal.rig.compose_height,
                    cal.rig.resize_input ? " (resized per frame)" : "", cal.rig.compose_warp_scale, cal.pano_roi.width, cal.pano_roi.height, cal.blend_width, g.num_bands);
            for (double gn : cal.gains) fprintf(stderr, " %.3f", gn);
            fprintf(stderr, "\n");
        } else {
            // BASELINE rig: the reference's rig model at full resolution, spherical warper with the full circle on out_w columns
            cal.rig = msshim::calibrateCameras(o.views, o.w, o.h, o.hfov, -1.0, 0.01, -1.0);
            comp_owner.reset(new msshim::Compositor(o.views, o.w, o.h, MS_PROJ_SPHERICAL, (float)(o.out_w / (2.0 * M_PI)), o.bands, o.cpw, o.out_w, o.out_h, 1, o.cpw ? o.update_mask : 0));
            for (int i = 0; i < o.views; ++i) {
                comp_owner->setCamera(i, cal.rig.K_compose[i], cal.rig.R[i]);
                comp_owner->setGain(i, 1.0 + 0.02 * (i - (o.views - 1) / 2.0));
            }
            comp_owner->buildMaps();
            comp_owner->buildMasks(true);
            comp_owner->init_gpu();
        }
        msshim::Compositor &comp = *comp_owner;
        if (o.solve_mesh && o.reference_calib && cal.rig.resize_input) { fprintf(stderr, "stitch_app: --solve-mesh with a resized compose scale is not wired up (use --compose-megapix -1)\n"); return 2; }
        const float warp_scale = o.reference_calib ? cal.rig.compose_warp_scale : (float)(o.out_w / (2.0 * M_PI));
        const bool resize_in = o.reference_calib && cal.rig.resize_input;
        const int cw = o.reference_calib ? cal.rig.compose_width : o.w, ch = o.reference_calib ? cal.rig.compose_height : o.h;
        hipStream_t stitch_stream, recal_stream;
        HIPCHECK(hipStreamCreateWithFlags(&stitch_stream, hipStreamNonBlocking));
        HIPCHECK(hipStreamCreateWithFlags(&recal_stream, hipStreamNonBlocking));
        if (o.cpw)
            for (int i = 0; i < o.views; ++i) {
                const ms_view_geom g = comp.viewGeom(i);
                std::vector<float> mx, my;
                make_mesh(g.roi.width, g.roi.height, 10, 10, 0.1 * i, 6.0, mx, my);
                comp.convertMeshToMap(i, mx.data(), my.data(), 10, 10, recal_stream);
            }

        // ---- capture side: the newest frame of every camera, in pinned memory ----------------------------------------
        Lockable<std::vector<HostFrame>> imgs;
        imgs.v.resize(o.views);
        // the cameras' frames sit back to back in ONE pinned block (view i at i * host_frame_bytes): a frame set then crosses PCIe as one copy -- six separate
        // 3 MB NV12 copies per frame cost the link 10 % in per-copy overhead (2 420 frames/s against the 2 670 the bytes allow; round 4)
        const size_t host_frame_bytes = o.nv12 ? (size_t)o.w * o.h * 3 / 2 : (size_t)o.w * o.h * 3;
        unsigned char *host_slab = nullptr;
        HIPCHECK(hipHostMalloc((void **)&host_slab, host_frame_bytes * o.views, hipHostMallocDefault));
        for (int i = 0; i < o.views; ++i) {
            HostFrame &f = imgs.v[i];
            f.w = o.w; f.h = o.h;
            f.p = host_slab + (size_t)i * host_frame_bytes;
            if (o.nv12) synth_nv12(f.p, o.w, o.h, i);          // (h * 3/2 rows of w bytes)
            else if (o.solve_mesh) synth_scene_frame(f.p, o.w, o.h, i, o.views, o.hfov, warp_scale, (i % 2 ? 1.0 : -1.0) * std::max(1.5, warp_scale / 200.0));   // odd / even cameras see the scene +-3 px apart
            else synth_frame(f.p, o.w, o.h, i);
            f.seq = 0;
        }
        std::atomic<bool> running{true};
        std::thread capture([&] {                       // "cameras": bump the sequence number of the frames at their own pace
            while (running.load()) {
                { std::lock_guard<std::mutex> lk(imgs.mu); for (auto &f : imgs.v) ++f.seq; }
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
        });

        // ---- stitch_one + results queue + consume -----------------------------------------------------------------
        // The uploaded frames are double-buffered and travel on their own stream: the H2D copies of frame t+1 overlap the stitch of frame t (the reference
        // uploads and stitches on one stream per view, timed.cpp:64-68).  up_done[b]: buffer b is filled; buf_free[b]: the stitch that read it is done.
        std::vector<DevMat> full_imgs_b[2], nv12_imgs_b[2];
        hipStream_t upload_stream;
        HIPCHECK(hipStreamCreateWithFlags(&upload_stream, hipStreamNonBlocking));
        hipEvent_t up_done[2], buf_free[2];
        bool buf_used[2] = {false, false};
        for (int b = 0; b < 2; ++b) {
            full_imgs_b[b].resize(o.views);
            if (!o.nv12) {     // BGR cameras: the same one-block layout (rows of w * 3 bytes), one copy per frame set
                unsigned char *slab = nullptr;
                HIPCHECK(hipMalloc((void **)&slab, host_frame_bytes * o.views));
                for (int i = 0; i < o.views; ++i) {
                    DevMat &m = full_imgs_b[b][i];
                    m.rows = o.h; m.cols = o.w; m.flags = MS_8UC3; m.step = (size_t)o.w * 3; m.data = slab + (size_t)i * host_frame_bytes;
                }
            } else
            for (auto &m : full_imgs_b[b]) m.create(o.h, o.w, MS_8UC3, 3);
            nv12_imgs_b[b].resize(o.nv12 ? o.views : 0);
            if (o.nv12) {      // one device block for the N NV12 frames of a set (rows of exactly `w` bytes, like the pinned block): filled by ONE copy per frame set
                unsigned char *slab = nullptr;
                HIPCHECK(hipMalloc((void **)&slab, host_frame_bytes * o.views));
                for (int i = 0; i < o.views; ++i) {
                    DevMat &m = nv12_imgs_b[b][i];
                    m.rows = o.h * 3 / 2; m.cols = o.w; m.flags = MS_8UC1; m.step = (size_t)o.w; m.data = slab + (size_t)i * host_frame_bytes;
                }
            }
            HIPCHECK(hipEventCreateWithFlags(&up_done[b], hipEventDisableTiming));
            HIPCHECK(hipEventCreateWithFlags(&buf_free[b], hipEventDisableTiming));
        }
        std::vector<DevMat> small_imgs(resize_in ? o.views : 0);
        for (auto &m : small_imgs) m.create(ch, cw, MS_8UC3, 3);
        const int RING = 4;
        std::vector<Slot> ring(RING);
        const ms_pano_geom pg = comp.panoGeom();
        const int ya = pg.canvas_y & ~1, yb = std::min(o.out_h, (pg.canvas_y + pg.dst_roi_final.height + 1) & ~1);
        for (auto &s : ring) {
            s.pano8u.create(o.out_h, o.out_w, MS_8UC3, 3);
            HIPCHECK(hipMemset2D(s.pano8u.data, s.pano8u.step, 0, (size_t)o.out_w * 3, o.out_h));
            if (o.i420) s.i420.create((yb - ya) * 3 / 2, o.out_w, MS_8UC1, 1, true);
            HIPCHECK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        }
        BlockingQueue<Slot *> results, free_slots;
        for (auto &s : ring) free_slots.push(&s);

        std::vector<unsigned char> last_pano((size_t)o.out_w * o.out_h * 3);
        unsigned long long checksum = 0;
        long long consumed = 0;
        DevMat encoder_frame;                           // consume(): the resized, letterboxed I420 frame the encoder gets (one buffer: the consumer is one thread)
        std::vector<unsigned char> encoder_host;
        hipStream_t consume_stream = nullptr;
        int consume_image_height = 0;
        unsigned long long consume_checksum = 0;
        if (o.consume_w > 0) {
            encoder_frame.create(o.consume_h * 3 / 2, o.consume_w, MS_8UC1, 1, true);
            encoder_host.resize((size_t)o.consume_w * o.consume_h * 3 / 2);
            HIPCHECK(hipStreamCreateWithFlags(&consume_stream, hipStreamNonBlocking));
        }
        std::thread consumer([&] {                      // consume(): wait for the frame, (I420,) bring it to the host side
            for (;;) {
                Slot *s = results.pop();
                if (!s) break;
                HIPCHECK(hipEventSynchronize(s->done));
                if (o.consume_w > 0) {                  // timed.cpp:251-316 without the download / CPU resize / CPU cvtColor: one kernel, then the encoder's copy
                    consume_image_height = msshim::consume_i420(s->pano8u, encoder_frame, o.consume_w, o.consume_h, true, (ms_stream)consume_stream);
                    HIPCHECK(hipMemcpyAsync(encoder_host.data(), encoder_frame.data, encoder_host.size(), hipMemcpyDeviceToHost, consume_stream));
                    HIPCHECK(hipStreamSynchronize(consume_stream));
                    if (s->seq == o.frames - 1) for (unsigned char b : encoder_host) consume_checksum = (consume_checksum ^ b) * 1099511628211ull;
                }
                if (s->seq == o.frames - 1) {           // keep the last panorama for the checksum / dump
                    HIPCHECK(hipMemcpy2D(last_pano.data(), (size_t)o.out_w * 3, s->pano8u.data, s->pano8u.step, (size_t)o.out_w * 3, o.out_h, hipMemcpyDeviceToHost));
                }
                ++consumed;
                free_slots.push(s);
            }
        });

        std::thread recalibrater;
        std::atomic<int> recalibrations{0};
        std::vector<int> view_round(o.views, 0);          // the round whose mesh each view holds (read after the thread is joined)
        std::atomic<int> solver_iterations{0};
        std::atomic<long long> total_keypoints{0}, total_matches{0}, total_inliers{0};
        std::atomic<float> max_disp{0.f};
        std::string recal_failure;
        if (o.solve_mesh)
            // recalibrateMesh (meshwarper.cpp:378-386) with the mesh actually solved: upload the current frames on the recalibration stream,
            // remap them (createMesh's images[idx]), optimise the mesh from matches and hand it to the compositor -- all while the stitcher
            // runs.  The feature front-end is not part of this build: the matches are synthetic (a parallax that drifts from round to round).
            recalibrater = std::thread([&] {
                try {
                    namespace ff = msshim::featurefinder;
                    msshim::MeshWarper mw(o.views, 10, 10, warp_scale, 1.0, 1.0);
                    mw.params().theta_rule = 1;
                    mw.params().global_dist = std::max(4, std::min(30, o.out_w / 128));    // GLOBAL_DIST = 30 is tuned to 1080p views; scaled for small rigs
                    std::vector<DevMat> recal_full(o.views), images(o.views);
                    std::vector<ms_view_geom> g(o.views);
                    for (int i = 0; i < o.views; ++i) {
                        g[i] = comp.viewGeom(i);
                        recal_full[i].create(o.h, o.w, MS_8UC3, 3);
                        images[i].create(g[i].roi.height, g[i].roi.width, MS_8UC3, 3);
                    }
                    int round = 1;
                    while (running.load()) {
                        std::this_thread::sleep_for(std::chrono::milliseconds(5));
                        {
                            std::lock_guard<std::mutex> lk(imgs.mu);
                            if (!o.nv12)
                                for (int i = 0; i < o.views; ++i)
                                    HIPCHECK(hipMemcpy2DAsync(recal_full[i].data, recal_full[i].step, imgs.v[i].p, (size_t)o.w * 3, (size_t)o.w * 3, o.h,
                                                              hipMemcpyHostToDevice, recal_stream));
                            HIPCHECK(hipStreamSynchronize(recal_stream));
                        }
                        for (int i = 0; i < o.views; ++i) {
                            ms_image xm, ym;
                            msshim::check(ms_get_maps(comp.raw(), i, &xm, &ym));
                            ms_image src = msshim::wrap(recal_full[i]), dst = msshim::wrap(images[i]);
                            msshim::check(ms_remap(&src, &xm, &ym, &dst, MS_INTER_LINEAR_FIXPT, MS_BORDER_CONSTANT, (ms_stream)recal_stream));   // cv::remap, meshwarper.cpp:72
                        }
                        // createMesh's front-end (meshwarper.cpp:82-119), all on the device: overlap masks, ORB, Hamming 2-NN + ratio test, RANSAC homographies
                        std::vector<DevMat> fmasks;
                        ff::featureMasks(images, fmasks, std::min(400, std::max(16, o.out_w / 10)), (ms_stream)recal_stream);
                        std::vector<ff::ImageFeatures<DevMat>> feats;
                        ff::findFeatures(images, fmasks, feats, (ms_stream)recal_stream);
                        std::vector<ff::MatchesInfo> pairwise(o.views);                      // NUM_IMAGES - 1 + wrapAround (calibration.cpp:284)
                        ff::matchFeatures(feats, pairwise, (ms_stream)recal_stream);
                        for (auto &m : fmasks) if (m.data) HIPCHECK(hipFree(m.data));
                        for (auto &f : feats) if (f.descriptors.data) HIPCHECK(hipFree(f.descriptors.data));
                        { size_t nk = 0, nm = 0, ni = 0; for (auto &f : feats) nk += f.keypoints.size(); for (auto &pm : pairwise) { nm += pm.matches.size(); ni += pm.num_inliers; }
                          total_keypoints += (long long)nk; total_matches += (long long)nm; total_inliers += (long long)ni; }
                        const ms_mesh_info info = mw.calibrateMeshWarp(comp, images, feats, pairwise, (ms_stream)recal_stream);
                        solver_iterations += info.iterations;
                        if (getenv("STITCH_APP_TRACE")) {
                            size_t nm = 0; for (auto &pm : pairwise) nm += pm.matches.size();
                            fprintf(stderr, "round %d: matches %zu rows %d nnz %d iterations %d error %.3g\n", round, nm, info.rows, info.nnz, info.iterations, info.error);
                        }
                        for (int i = 0; i < o.views; ++i) {
                            float dpx = 0.f;
                            msshim::check(ms_get_mesh_displacement(comp.raw(), i, &dpx));
                            if (dpx > max_disp.load()) max_disp.store(dpx);
                        }
                        ++round; ++recalibrations;
                    }
                } catch (const std::exception &e) { recal_failure = e.what(); }
            });
        else if (o.cpw)
            recalibrater = std::thread([&] {            // timed.cpp:414-463: new meshes while the stitcher keeps running
                int round = 1;
                while (running.load()) {
                    std::this_thread::sleep_for(std::chrono::milliseconds(5));
                    for (int i = 0; i < o.views && running.load(); ++i) {
                        const ms_view_geom g = comp.viewGeom(i);
                        std::vector<float> mx, my;
                        make_mesh(g.roi.width, g.roi.height, 10, 10, 0.1 * i + 0.37 * round, 6.0, mx, my);
                        comp.convertMeshToMap(i, mx.data(), my.data(), 10, 10, recal_stream);
                        if (o.update_mask > 0) comp.update_mask(i, (ms_stream)recal_stream);      // timed.cpp:598-605
                        view_round[i] = round;
                    }
                    ++round; ++recalibrations;
                }
            });

        const auto t0 = std::chrono::steady_clock::now();
        std::string failure;
        try {
        for (int t = 0; t < o.frames; ++t) {            // main loop: stitch_one per frame
            Slot *s = free_slots.pop();
            const int ib = o.upload ? (t & 1) : 0;
            std::vector<DevMat> &full_imgs = full_imgs_b[ib], &nv12_imgs = nv12_imgs_b[ib];
            if (o.upload || t == 0) {
                std::lock_guard<std::mutex> lk(imgs.mu);                                  // imgs.lock() ... imgs.unlock()
                if (buf_used[ib]) HIPCHECK(hipStreamWaitEvent(upload_stream, buf_free[ib], 0));     // the stitch of frame t-2 has read this buffer
                if (o.nv12)               // half the PCIe bytes: upload NV12 (all cameras in one copy), convert on the device / in the warp
                    HIPCHECK(hipMemcpyAsync(nv12_imgs[0].data, host_slab, host_frame_bytes * o.views, hipMemcpyHostToDevice, upload_stream));
                else                      // GpuMat::upload(Mat, stream) of every camera (timed.cpp:68), as one copy
                    HIPCHECK(hipMemcpyAsync(full_imgs[0].data, host_slab, host_frame_bytes * o.views, hipMemcpyHostToDevice, upload_stream));
                HIPCHECK(hipEventRecord(up_done[ib], upload_stream));
                HIPCHECK(hipStreamWaitEvent(stitch_stream, up_done[ib], 0));
                if (o.nv12 && !(o.nv12_direct && !resize_in)) {             // cvtColor(YUV2BGR_NV12) of all cameras in one launch (also with --nv12-direct when the frames are resized first: the resize works on BGR) (the reference: per camera, on the CPU, networking.cpp:45-47).
                                          // On the STITCH stream (round 4): the upload stream then carries nothing but the copies, so the PCIe link -- the limit of
                                          // this path -- never waits for a kernel; the conversion (one latency-bound launch) rides in front of the frame's stitch
                    std::vector<ms_image> a(o.views), d(o.views);
                    for (int i = 0; i < o.views; ++i) { a[i] = msshim::wrap(nv12_imgs[i]); d[i] = msshim::wrap(full_imgs[i]); }
                    msshim::check(ms_nv12_to_bgr_batch(a.data(), d.data(), o.views, (ms_stream)stitch_stream));
                }
            }
            if (resize_in) {                        // timed.cpp:75-85: cuda::resize(full_imgs[i], resized, Size(), compose_scale, compose_scale)
                msshim::cuda::resize(full_imgs, small_imgs, cal.rig.compose_scale, cal.rig.compose_scale, (ms_stream)stitch_stream);      // all views, one launch
                comp.stitch_one(small_imgs, &s->pano8u, (DevMat *)nullptr, (ms_stream)stitch_stream);
            } else if (o.nv12_direct)      // (resize_in is false here)
                comp.stitch_one_nv12(nv12_imgs, &s->pano8u, (DevMat *)nullptr, (ms_stream)stitch_stream);      // the warp converts each tap itself: no BGR frames on the device at all
            else
                comp.stitch_one(full_imgs, &s->pano8u, (DevMat *)nullptr, (ms_stream)stitch_stream);
            if (o.i420) {
                ms_image rows{s->pano8u.data + (size_t)ya * s->pano8u.step, s->pano8u.step, o.out_w, yb - ya, MS_8UC3};
                ms_image dst = msshim::wrap(s->i420);
                msshim::check(ms_bgr_to_i420(&rows, &dst, (ms_stream)stitch_stream));       // cvtColor(BGR2YUV_I420), timed.cpp:308-316
            }
            s->seq = t;
            HIPCHECK(hipEventRecord(s->done, stitch_stream));
            HIPCHECK(hipEventRecord(buf_free[ib], stitch_stream)); buf_used[ib] = true;
            results.push(s);
        }
        } catch (const msshim::Error &e) { failure = e.what(); }        // threads must be joined before the error is reported
        results.push(nullptr);
        consumer.join();
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        running.store(false);
        capture.join();
        if (recalibrater.joinable()) recalibrater.join();
        HIPCHECK(hipDeviceSynchronize());
        if (!failure.empty()) { fprintf(stderr, "stitch_app: %s\n", failure.c_str()); return 1; }
        if (!recal_failure.empty()) { fprintf(stderr, "stitch_app (recalibration): %s\n", recal_failure.c_str()); return 1; }

        // --update-mask self-check: whatever interleaving of stitches, mesh swaps and enqueue-only mask updates happened above, the tables the context ended
        // up with must be exactly what a fresh context gets from the same final meshes with the synchronous update_mask (one more frame through both)
        int selfcheck = -1;
        if (o.update_mask > 0 && !o.solve_mesh && !o.reference_calib) {
            msshim::Compositor ref(o.views, o.w, o.h, MS_PROJ_SPHERICAL, (float)(o.out_w / (2.0 * M_PI)), o.bands, true, o.out_w, o.out_h, 1, 0);
            for (int i = 0; i < o.views; ++i) {
                ref.setCamera(i, cal.rig.K_compose[i], cal.rig.R[i]);
                ref.setGain(i, 1.0 + 0.02 * (i - (o.views - 1) / 2.0));
            }
            ref.buildMaps(); ref.buildMasks(true); ref.init_gpu();
            std::vector<DevMat> views(o.views);
            std::vector<unsigned char> host((size_t)o.w * o.h * 3);
            for (int i = 0; i < o.views; ++i) {
                const ms_view_geom g = ref.viewGeom(i);
                std::vector<float> mx, my;
                make_mesh(g.roi.width, g.roi.height, 10, 10, 0.1 * i + (view_round[i] ? 0.37 * view_round[i] : 0.0), 6.0, mx, my);
                ref.convertMeshToMap(i, mx.data(), my.data(), 10, 10, nullptr);
                views[i].create(o.h, o.w, MS_8UC3, 3);
                synth_frame(host.data(), o.w, o.h, (i + 3) % o.views);
                HIPCHECK(hipMemcpy2D(views[i].data, views[i].step, host.data(), (size_t)o.w * 3, (size_t)o.w * 3, o.h, hipMemcpyHostToDevice));
            }
            for (int i = 0; i < o.views; ++i)
                if (view_round[i]) ref.update_mask(i, nullptr);          // (round 0 = the start-up meshes: no mask update was issued for them)
            DevMat a, b;
            a.create(o.out_h, o.out_w, MS_8UC3, 3); b.create(o.out_h, o.out_w, MS_8UC3, 3);
            HIPCHECK(hipMemset2D(a.data, a.step, 0, (size_t)o.out_w * 3, o.out_h)); HIPCHECK(hipMemset2D(b.data, b.step, 0, (size_t)o.out_w * 3, o.out_h));
            comp.stitch_one(views, &a, (DevMat *)nullptr, (ms_stream)stitch_stream);
            ref.stitch_one(views, &b, (DevMat *)nullptr, nullptr);
            HIPCHECK(hipDeviceSynchronize());
            std::vector<unsigned char> ha((size_t)o.out_w * o.out_h * 3), hb(ha.size());
            HIPCHECK(hipMemcpy2D(ha.data(), (size_t)o.out_w * 3, a.data, a.step, (size_t)o.out_w * 3, o.out_h, hipMemcpyDeviceToHost));
            HIPCHECK(hipMemcpy2D(hb.data(), (size_t)o.out_w * 3, b.data, b.step, (size_t)o.out_w * 3, o.out_h, hipMemcpyDeviceToHost));
            selfcheck = ha == hb ? 1 : 0;
            for (auto &m : views) HIPCHECK(hipFree(m.data));
            HIPCHECK(hipFree(a.data)); HIPCHECK(hipFree(b.data));
        }
        for (unsigned char b : last_pano) checksum = (checksum ^ b) * 1099511628211ull;   // FNV-1a over the last 8U panorama
        if (!o.dump.empty()) {
            FILE *f = fopen(o.dump.c_str(), "wb");
            if (!f || fwrite(last_pano.data(), 1, last_pano.size(), f) != last_pano.size()) { fprintf(stderr, "cannot write %s\n", o.dump.c_str()); return 2; }
            fclose(f);
        }
        printf("{\"app\": \"stitch_app\", \"views\": %d, \"src\": \"%dx%d\", \"out\": \"%dx%d\", \"bands\": %d, \"cpw\": %s, \"i420\": %s, \"nv12\": %s, \"nv12_direct\": %s, \"upload\": %s, "
               "\"frames\": %lld, \"seconds\": %.4f, \"frames_per_s\": %.1f, \"recalibrations\": %d, \"mesh_solver_iterations\": %d, \"max_mesh_displacement_px\": %.2f, "
               "\"orb_keypoints\": %lld, \"ratio_matches\": %lld, \"ransac_inliers\": %lld, \"update_mask_margin\": %d, \"update_mask_equals_sync_rebuild\": %s, \"consume_image_height\": %d, \"consume_checksum\": \"%016llx\", \"checksum\": \"%016llx\"}\n",
               o.views, o.w, o.h, o.out_w, o.out_h, pg.num_bands, o.cpw ? "true" : "false", o.i420 ? "true" : "false", o.nv12 ? "true" : "false", o.nv12_direct ? "true" : "false", o.upload ? "true" : "false",
               consumed, secs, consumed / secs, recalibrations.load(), solver_iterations.load(), (double)max_disp.load(),
               total_keypoints.load(), total_matches.load(), total_inliers.load(), o.update_mask, selfcheck < 0 ? "null" : (selfcheck ? "true" : "false"), consume_image_height, consume_checksum, checksum);
    } catch (const msshim::Error &e) {
        fprintf(stderr, "stitch_app: msstitch error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
