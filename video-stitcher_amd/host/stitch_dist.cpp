// stitch_dist.cpp -- the multi-GPU host pipeline over the C-ABI (include/ms_stitch.h + include/ms_dist.h): one thread per MI355X, each with its
// own compositor context and its own rank of an ms_dist communicator (RCCL over xGMI; the shared-memory host transport when ranks share a GPU).
//
//   SURVEY.md 8(e) / BASELINE configs[3]   frame-parallel: frames are independent given the calibration tables, every rank holds a replica of the
//                                          static tables, global frame t of batch k goes to group (t mod groups); the finished panoramas travel
//                                          as planar I420 slabs (consume()'s encoder input, APP/timed.cpp:308-316) to the sink rank 0
//   BASELINE configs[4]                    --col-shards S: a group = S ranks that composite S column windows of the SAME frames (96-px halo,
//                                          ms_config.col_shards) and send their window of the I420 planes to the group's first rank;
//                                          G = groups x S ranks, e.g. 8 GPUs = 4 frames in flight x 2 GPUs per frame
//   recalibration (APP/timed.cpp:414-463,  --cpw --recalib-every K: a recalibration thread on rank 0 produces new N x M meshes; rank 0 announces
//   meshwarper.cpp:879-884)                them one batch ahead with the global frame index from which they apply (ms_dist_mesh_exchange =
//                                          ncclBroadcast), every rank runs convertMeshesToMap itself (ms_set_mesh) when it reaches that frame
//
// The reference is single-device (timed.cpp:496 cuda::setDevice(0)); this is its main loop instantiated once per GPU.  Whatever G, S and the
// transport are, the I420 frames arriving on the sink are bit-identical to the single-GPU run: `checksum_all` (FNV-1a over the per-frame
// checksums in display order) is what tests/test_ms_dist_gpu.py compares.
//
// Usage: stitch_dist [--gpus G] [--col-shards S] [--share-gpu] [--transport auto|rccl|host] [--frames T] [--batch F] [--views 6] [--size WxH]
//                    [--out WxH] [--hfov 90] [--bands 5] [--cpw] [--recalib-every K] [--mesh NxM] [--no-checksum] [--tables-from-rank0] [--rccl-lib FILE]
// Prints one JSON line (rank 0): frames/s of the whole job, what the communicator saw (transport, nranks, devices, PCI ids), the checksums.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include "../../include/ms_dist.h"

namespace {

struct Fail : std::runtime_error { using std::runtime_error::runtime_error; };
void chk(int code, const char *what) { if (code < 0) throw Fail(std::string(what) + ": " + ms_last_error()); }
void hchk(hipError_t e, const char *what) { if (e != hipSuccess) throw Fail(std::string(what) + ": " + hipGetErrorString(e)); }
#define MSC(x) chk((x), #x)
#define HIPC(x) hchk((x), #x)

struct Options {
    int gpus = 1, col_shards = 1, frames = 64, batch = 4, views = 6, w = 1920, h = 1080, out_w = 3840, out_h = 1920, bands = 5;
    int recalib_every = 0, mesh_rows = 10, mesh_cols = 10, transport = MS_DIST_AUTO;
    double hfov = 90.0;
    bool share_gpu = false, cpw = false, checksum = true;
    bool frame_sums = false;          // --frame-sums: print every frame's checksum on stderr (diagnostic)
    bool tables_from_rank0 = false;   // --tables-from-rank0: only rank 0 calibrates; the others build their context from its table blob (ms_save_tables -> ms_dist_broadcast -> ms_load_tables)
    std::string rccl_lib;             // --rccl-lib FILE: ms_dist_set_rccl_library (a deployment's own RCCL, or the loopback implementation the tests substitute)
};

// the pattern of video-stitcher_amd/synth.py (noise off), `variant` shifts the phase so that consecutive frames differ
void synth_frame(unsigned char *dst, int w, int h, int view, int variant)
{
    const double two_pi = 2.0 * M_PI;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const double phase = x / 97.0 + y / 61.0 + (view + 2 * variant) / 7.0;
            const double chk = 40.0 * ((((x + 8 * variant) / 32) + (y / 32)) & 1);
            for (int c = 0; c < 3; ++c) {
                double v = std::nearbyint(128.0 + 60.0 * std::sin(two_pi * (phase + c / 3.0)) + chk);
                dst[((size_t)y * w + x) * 3 + c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
}
// a smooth synthetic CPW mesh (identity + amp sin(2 pi u + phase) sin(pi v)); round r of the recalibration thread = another phase
void make_mesh(int aw, int ah, int N, int M, double phase, double amp, float *mx, float *my)
{
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < M; ++j) {
            const double u = (double)j / (M - 1), v = (double)i / (N - 1), d = amp * std::sin(2.0 * M_PI * u + phase) * std::sin(M_PI * v);
            mx[(size_t)i * M + j] = (float)(u * (aw - 1) + d);
            my[(size_t)i * M + j] = (float)(v * (ah - 1) + 0.5 * d);
        }
}
unsigned long long fnv(const unsigned char *p, size_t n, unsigned long long h = 1469598103934665603ull)
{
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

struct MeshSet { int round; std::vector<float> x, y; };
template <class T> class BlockingQueue {       // blockingqueue.h of the reference: unbounded push, blocking pop
public:
    void push(T v) { { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(v)); } cv_.notify_one(); }
    T pop() { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return !q_.empty(); }); T v = std::move(q_.front()); q_.pop_front(); return v; }
    size_t size() { std::lock_guard<std::mutex> lk(mu_); return q_.size(); }
private:
    std::mutex mu_; std::condition_variable cv_; std::deque<T> q_;
};

struct Shared {                                // what the rank threads share: the id, failure reports, rank 0's results
    unsigned char id[MS_DIST_ID_BYTES];
    std::mutex mu;
    std::string failure;
    std::vector<unsigned long long> frame_sums;
    double seconds = 0;
    ms_dist_info info{};
    int bands = 0, i_rows = 0, recalibrations = 0;
    std::vector<int> views_read;
    // where each rank's time went (the first real multi-GPU run should explain itself): GPU time of the stitch kernels (events on the
    // stitch stream), time of the gather on the communication stream (events; on the host transport this is host-staged copies), host time blocked in the
    // mesh exchange / the column-shard exchange / the sink's consume loop
    struct RankTimes { double stitch_gpu_ms = 0, gather_stream_ms = 0, mesh_exchange_host_ms = 0, shard_exchange_host_ms = 0, consume_host_ms = 0, wall_ms = 0; };
    std::vector<RankTimes> times;
};

void rank_main(const Options &o, int rank, Shared &sh)
{
    int ndev = 0;
    HIPC(hipGetDeviceCount(&ndev));
    const int dev = o.share_gpu ? 0 : rank % ndev;
    HIPC(hipSetDevice(dev));
    const int S = o.col_shards, groups = o.gpus / S, group = rank / S, shard = rank % S, leader = group * S, F = o.batch, N = o.views;
    hipStream_t st;
    HIPC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

    // ---- stitch_calib on this rank: a replica of the static tables ----------------------------------------------------
    ms_config c{};
    c.struct_size = (unsigned)sizeof(ms_config);
    c.num_views = N; c.src_width = o.w; c.src_height = o.h; c.projection = MS_PROJ_SPHERICAL; c.warp_scale = (float)(o.out_w / (2.0 * M_PI));
    c.num_bands = o.bands; c.enable_cpw = o.cpw; c.out_width = o.out_w; c.out_height = o.out_h; c.max_frames = F;
    c.col_shards = S; c.col_shard_index = shard;
    ms_ctx *ctx = nullptr;
    // the communicator comes first when the tables travel through it
    ms_dist *dist = nullptr;
    if (o.tables_from_rank0) MSC(ms_dist_create(&dist, rank, o.gpus, sh.id, dev));
    if (!o.tables_from_rank0 || rank == 0) {
        MSC(ms_create(&c, &ctx));
        ms_rig_params rp{N, o.w, o.h, o.hfov, -1.0, 0.01, -1.0};
        ms_rig rig;
        MSC(ms_calibrate_cameras(&rp, &rig));
        for (int i = 0; i < N; ++i) { MSC(ms_set_camera(ctx, i, rig.K_compose[i], rig.R[i])); MSC(ms_set_gain(ctx, i, 1.0 + 0.02 * (i - (N - 1) / 2.0))); }
        MSC(ms_build_maps(ctx, st)); MSC(ms_build_masks(ctx, 1, st)); MSC(ms_init_blender(ctx, st));
    }
    if (o.tables_from_rank0) {       // timed.cpp:553 calibrates at every start; here ONE rank does and every other rank rebuilds identical tables from its blob
        unsigned long long nbytes = 0;
        std::vector<unsigned char> blob;
        if (rank == 0) { size_t n = 0; MSC(ms_save_tables(ctx, nullptr, 0, &n)); blob.resize(n); MSC(ms_save_tables(ctx, blob.data(), n, &n)); nbytes = n; }
        if (o.gpus > 1) MSC(ms_dist_broadcast(dist, &nbytes, sizeof(nbytes), 0, MS_DIST_MEM_HOST, st));
        blob.resize((size_t)nbytes);
        if (o.gpus > 1) MSC(ms_dist_broadcast(dist, blob.data(), blob.size(), 0, MS_DIST_MEM_HOST, st));
        if (rank != 0) MSC(ms_load_tables(blob.data(), blob.size(), &ctx, st));
    }
    ms_pano_geom pg;
    MSC(ms_get_pano_geom(ctx, &pg));
    int i_y0 = 0, i_rows = 0, win_b = 0, win_e = 0;
    MSC(ms_get_i420_rows(ctx, &i_y0, &i_rows));
    MSC(ms_get_col_window(ctx, &win_b, &win_e));
    unsigned need = 0;
    MSC(ms_get_needed_views(ctx, &need));
    std::vector<ms_view_geom> vg(N);
    for (int i = 0; i < N; ++i) MSC(ms_get_view_geom(ctx, i, &vg[i]));
    const size_t per_mesh = (size_t)o.mesh_rows * o.mesh_cols;
    if (o.cpw) {                               // the start-up meshes (round 0) are part of the calibration every rank runs itself
        std::vector<float> mx(per_mesh), my(per_mesh);
        for (int i = 0; i < N; ++i) {
            make_mesh(vg[i].roi.width, vg[i].roi.height, o.mesh_rows, o.mesh_cols, 0.1 * i, 6.0, mx.data(), my.data());
            MSC(ms_set_mesh(ctx, i, mx.data(), my.data(), o.mesh_rows, o.mesh_cols, st));
        }
    }

    // ---- the communicator ----------------------------------------------------------------------------------------------------------
    if (!dist) MSC(ms_dist_create(&dist, rank, o.gpus, sh.id, dev));
    ms_dist_info info;
    MSC(ms_dist_get_info(dist, &info));

    // ---- source frames: a pool of 4 frame sets resident in HBM (only the views this rank's window reads) ----------------------------
    const int POOL = 4;
    std::vector<unsigned char *> src((size_t)POOL * N, nullptr);
    {
        std::vector<unsigned char> host((size_t)o.w * o.h * 3);
        for (int p = 0; p < POOL; ++p)
            for (int i = 0; i < N; ++i) {
                if (!((need >> i) & 1u)) continue;
                synth_frame(host.data(), o.w, o.h, i, p);
                HIPC(hipMalloc((void **)&src[(size_t)p * N + i], host.size()));
                HIPC(hipMemcpy(src[(size_t)p * N + i], host.data(), host.size(), hipMemcpyHostToDevice));
            }
    }
    // ---- outputs: F planar I420 frames of the panorama's canvas rows; the sink has one set per group, a leader receive space for its shards
    const size_t y_bytes = (size_t)o.out_w * i_rows, frame_bytes = y_bytes * 3 / 2, slab_bytes = frame_bytes * F;
    auto black = [&](unsigned char *p) {
        for (int f = 0; f < F; ++f) { HIPC(hipMemsetAsync(p + f * frame_bytes, 16, y_bytes, st)); HIPC(hipMemsetAsync(p + f * frame_bytes + y_bytes, 128, y_bytes / 2, st)); }
    };
    // double-buffered: the slabs of batch k travel to the sink on a stream of their own while batch k + 1 is stitched
    unsigned char *mine2[2] = {nullptr, nullptr};
    std::vector<unsigned char *> from_group2[2];                    // sink: slabs of the other groups' leaders
    hipStream_t cs;
    HIPC(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipEvent_t stitched[2], sent[2];
    bool sent_used[2] = {false, false};
    for (int b = 0; b < 2; ++b) {
        HIPC(hipMalloc((void **)&mine2[b], slab_bytes));
        black(mine2[b]);
        from_group2[b].assign(groups, nullptr);
        if (rank == 0) for (int g = 1; g < groups; ++g) HIPC(hipMalloc((void **)&from_group2[b][g], slab_bytes));
        HIPC(hipEventCreateWithFlags(&stitched[b], hipEventDisableTiming));
        HIPC(hipEventCreateWithFlags(&sent[b], hipEventDisableTiming));
    }
    // column windows of every shard (a pure function of the panorama width: ms_get_col_window documents the rule)
    const int fw = pg.dst_roi_final.width;
    auto bound = [&](int i) { return i <= 0 ? 0 : (i >= S ? fw : (int)((long long)i * fw / S) / 16 * 16); };
    // chroma belongs to the EVEN canvas columns (top-left pixel of each 2 x 2 block, timed.cpp:308-316): a window [b, e) of panorama columns owns the
    // chroma samples of the even canvas columns inside it -- canvas_x may be odd, so these are columns ceil(cb / 2) .. ceil(ce / 2) - 1 of the U / V planes
    auto chroma_span = [&](int b, int e, int *x0) { const int cb = b + pg.canvas_x, ce = e + pg.canvas_x; *x0 = (cb + 1) >> 1; return ((ce + 1) >> 1) - ((cb + 1) >> 1); };
    auto win_bytes = [&](int k) { int x0; const int cw2 = chroma_span(bound(k), bound(k + 1), &x0); return ((size_t)(bound(k + 1) - bound(k)) * i_rows + (size_t)cw2 * i_rows) * F; };
    if (bound(shard) != win_b || bound(shard + 1) != win_e) throw Fail("column window rule out of step with ms_get_col_window");
    std::vector<unsigned char *> from_shard(S, nullptr);            // leader: packed windows of its shards; shard: its own packed window
    if (S > 1) {
        if (rank == leader) { for (int k = 1; k < S; ++k) HIPC(hipMalloc((void **)&from_shard[k], win_bytes(k))); }
        else HIPC(hipMalloc((void **)&from_shard[shard], win_bytes(shard)));
    }
    // pack / unpack the window [b, e) of the Y, U, V planes of F frames (panorama columns -> canvas columns + canvas_x)
    auto move_window = [&](unsigned char *frames, unsigned char *packed, int b, int e, bool pack) {
        const int cb = b + pg.canvas_x, wy = e - b;
        int cx0;
        const int cw2 = chroma_span(b, e, &cx0);
        unsigned char *q = packed;
        for (int f = 0; f < F; ++f) {
            unsigned char *Y = frames + f * frame_bytes, *U = Y + y_bytes, *V = U + y_bytes / 4;
            struct { unsigned char *p; int pitch, x, w, rows; } pl[3] = {{Y, o.out_w, cb, wy, i_rows}, {U, o.out_w / 2, cx0, cw2, i_rows / 2}, {V, o.out_w / 2, cx0, cw2, i_rows / 2}};
            for (auto &P : pl) {
                if (P.w <= 0) continue;
                if (pack) HIPC(hipMemcpy2DAsync(q, P.w, P.p + P.x, P.pitch, P.w, P.rows, hipMemcpyDeviceToDevice, st));
                else HIPC(hipMemcpy2DAsync(P.p + P.x, P.pitch, q, P.w, P.w, P.rows, hipMemcpyDeviceToDevice, st));
                q += (size_t)P.w * P.rows;
            }
        }
    };

    // ---- recalibration thread on rank 0 (timed.cpp:414-463): meshes of round r, produced ahead of their use -----------------------------
    BlockingQueue<MeshSet> solved;
    std::atomic<bool> running{true};
    std::thread recalibrater;
    const long long batch_frames = (long long)groups * F, n_batches = (o.frames + batch_frames - 1) / batch_frames;
    const int n_rounds = (o.cpw && o.recalib_every > 0) ? (int)((n_batches * batch_frames - 1) / o.recalib_every) : 0;
    if (rank == 0 && n_rounds > 0)
        recalibrater = std::thread([&] {
            for (int r = 1; r <= n_rounds && running.load(); ++r) {
                MeshSet m{r, std::vector<float>(per_mesh * N), std::vector<float>(per_mesh * N)};
                for (int i = 0; i < N; ++i)
                    make_mesh(vg[i].roi.width, vg[i].roi.height, o.mesh_rows, o.mesh_cols, 0.1 * i + 0.37 * r, 6.0, m.x.data() + i * per_mesh, m.y.data() + i * per_mesh);
                while (solved.size() >= 2 && running.load()) std::this_thread::sleep_for(std::chrono::microseconds(200));
                solved.push(std::move(m));
            }
        });
    std::vector<float> pend_x(per_mesh * N), pend_y(per_mesh * N);
    ms_dist_mesh_update pending{0, 0, 0, 0, 0, pend_x.data(), pend_y.data()};
    bool have_pending = false;
    int applied_rounds = 0;

    Shared::RankTimes rt;
    // per-batch timing events: a ring of RING (begin, end) pairs per timer, re-used; a pair's elapsed time is added up when its slot comes round again (and at the end), so
    // a run of any length holds 4 x RING events
    constexpr int RING = 8;
    struct Timer {
        hipEvent_t e0[RING], e1[RING];
        bool open[RING] = {};
        int n = 0;
        double *sum = nullptr;
        void retire(int i) { if (open[i]) { HIPC(hipEventSynchronize(e1[i])); float t = 0; HIPC(hipEventElapsedTime(&t, e0[i], e1[i])); *sum += t; open[i] = false; } }
        void begin(hipStream_t s) { const int i = n % RING; retire(i); HIPC(hipEventRecord(e0[i], s)); }
        void end(hipStream_t s) { const int i = n % RING; HIPC(hipEventRecord(e1[i], s)); open[i] = true; ++n; }
    } t_stitch, t_gather;
    t_stitch.sum = &rt.stitch_gpu_ms; t_gather.sum = &rt.gather_stream_ms;
    for (Timer *t : {&t_stitch, &t_gather}) for (int i = 0; i < RING; ++i) { HIPC(hipEventCreate(&t->e0[i])); HIPC(hipEventCreate(&t->e1[i])); }
    auto tick = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    std::vector<unsigned char> host_frame(rank == 0 && o.checksum ? frame_bytes : 0);
    std::vector<unsigned long long> sums;
    HIPC(hipStreamSynchronize(st));
    MSC(ms_dist_barrier(dist, st));
    const auto t0 = std::chrono::steady_clock::now();
    for (long long k = 0; k < n_batches; ++k) {
        const long long first = k * batch_frames;                  // global index of this batch's first frame
        const auto t_mesh = tick();
        if (o.cpw && o.recalib_every > 0) {
            // 1. a pending update whose swap frame has come is applied by every rank at the same batch boundary
            if (have_pending) {
                int applied = 0;
                MSC(ms_dist_apply_meshes(ctx, &pending, first, &applied, st));
                if (applied) { have_pending = false; ++applied_rounds; }
            }
            // 2. rank 0 announces the next round one batch ahead: swap_frame = the first frame of the NEXT batch when that is a recalibration point
            const long long next_first = first + batch_frames;
            ms_dist_mesh_update upd{}, *up = nullptr;
            MeshSet m;
            if (rank == 0 && next_first % o.recalib_every == 0 && next_first < n_batches * batch_frames) {
                m = solved.pop();                                  // (blocks until the recalibration thread has this round: keeps the run deterministic)
                upd = ms_dist_mesh_update{next_first, m.round, N, o.mesh_rows, o.mesh_cols, m.x.data(), m.y.data()};
                up = &upd;
            }
            int have = 0;
            MSC(ms_dist_mesh_exchange(dist, 0, up, &pending, per_mesh * N, &have, st));
            if (have) { if (have_pending) throw Fail("a second mesh update arrived before the first was applied"); have_pending = true; }
        }
        rt.mesh_exchange_host_ms += ms_since(t_mesh);
        // 3. this group's F frames of the batch: t = first + j * groups + group
        const int b = (int)(k & 1);
        unsigned char *mine = mine2[b];
        std::vector<unsigned char *> &from_group = from_group2[b];
        if (sent_used[b]) HIPC(hipStreamWaitEvent(st, sent[b], 0));      // the slab this buffer held two batches ago has left
        std::vector<ms_image> views((size_t)F * N), outs(F);
        for (int j = 0; j < F; ++j) {
            const long long t = first + (long long)j * groups + group;
            for (int i = 0; i < N; ++i) {
                unsigned char *p = src[(size_t)(t % POOL) * N + i];
                views[(size_t)j * N + i] = p ? ms_image{p, (size_t)o.w * 3, o.w, o.h, MS_8UC3} : ms_image{nullptr, 0, 0, 0, 0};
            }
            outs[j] = ms_image{mine + j * frame_bytes, (size_t)o.out_w, o.out_w, i_rows * 3 / 2, MS_8UC1};
        }
        t_stitch.begin(st);
        MSC(ms_stitch_i420(ctx, F, views.data(), outs.data(), st));
        t_stitch.end(st);
        const auto t_shard = tick();
        // 4. column shards: windows to the group's first rank
        if (S > 1) {
            if (rank != leader) {
                move_window(mine, from_shard[shard], win_b, win_e, true);
                MSC(ms_dist_send(dist, from_shard[shard], win_bytes(shard), leader, MS_DIST_MEM_DEVICE, st));
            } else {
                MSC(ms_dist_group_begin(dist));
                for (int s2 = 1; s2 < S; ++s2) MSC(ms_dist_recv(dist, from_shard[s2], win_bytes(s2), leader + s2, MS_DIST_MEM_DEVICE, st));
                MSC(ms_dist_group_end(dist));
                for (int s2 = 1; s2 < S; ++s2) move_window(mine, from_shard[s2], bound(s2), bound(s2 + 1), false);
            }
        }
        rt.shard_exchange_host_ms += ms_since(t_shard);
        // 5. frame-parallel gather: the leaders' slabs to the sink (rank 0), on the communication stream behind this batch's kernels
        HIPC(hipEventRecord(stitched[b], st));
        if (groups > 1 && (rank == 0 || rank == leader)) {
            HIPC(hipStreamWaitEvent(cs, stitched[b], 0));
            t_gather.begin(cs);
            if (rank == 0) {
                MSC(ms_dist_group_begin(dist));
                for (int g = 1; g < groups; ++g) MSC(ms_dist_recv(dist, from_group[g], slab_bytes, g * S, MS_DIST_MEM_DEVICE, cs));
                MSC(ms_dist_group_end(dist));
            } else
                MSC(ms_dist_send(dist, mine, slab_bytes, 0, MS_DIST_MEM_DEVICE, cs));
            HIPC(hipEventRecord(sent[b], cs));
            t_gather.end(cs);
            sent_used[b] = true;
        }
        const auto t_cons = tick();
        // 6. consume() on the sink: the frames of the batch in display order
        if (rank == 0 && o.checksum) {
            HIPC(hipStreamSynchronize(st));
            HIPC(hipStreamSynchronize(cs));
            for (int j = 0; j < F; ++j)
                for (int g = 0; g < groups; ++g) {
                    if (first + (long long)j * groups + g >= o.frames) continue;
                    const unsigned char *p = (g == 0 ? mine : from_group[g]) + j * frame_bytes;
                    HIPC(hipMemcpyAsync(host_frame.data(), p, frame_bytes, hipMemcpyDeviceToHost, cs));
                    HIPC(hipStreamSynchronize(cs));
                    sums.push_back(fnv(host_frame.data(), frame_bytes));
                }
        }
        rt.consume_host_ms += ms_since(t_cons);
    }
    HIPC(hipStreamSynchronize(st));
    HIPC(hipStreamSynchronize(cs));
    MSC(ms_dist_barrier(dist, st));
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    running.store(false);
    if (recalibrater.joinable()) recalibrater.join();
    rt.wall_ms = secs * 1e3;
    for (Timer *t : {&t_stitch, &t_gather}) for (int i = 0; i < RING; ++i) { t->retire(i); (void)hipEventDestroy(t->e0[i]); (void)hipEventDestroy(t->e1[i]); }
    if (rank == 0) {
        std::lock_guard<std::mutex> lk(sh.mu);
        sh.frame_sums = sums; sh.seconds = secs; sh.info = info; sh.bands = pg.num_bands; sh.i_rows = i_rows; sh.recalibrations = applied_rounds;
    }
    { std::lock_guard<std::mutex> lk(sh.mu); if ((int)sh.views_read.size() < o.gpus) sh.views_read.resize(o.gpus); sh.views_read[rank] = __builtin_popcount(need);
      if ((int)sh.times.size() < o.gpus) sh.times.resize(o.gpus); sh.times[rank] = rt; }
    ms_dist_destroy(dist);
    ms_destroy(ctx);
    for (unsigned char *p : src) if (p) (void)hipFree(p);
    for (int b = 0; b < 2; ++b) {
        for (unsigned char *p : from_group2[b]) if (p) (void)hipFree(p);
        (void)hipFree(mine2[b]);
        (void)hipEventDestroy(stitched[b]); (void)hipEventDestroy(sent[b]);
    }
    for (unsigned char *p : from_shard) if (p) (void)hipFree(p);
    (void)hipStreamDestroy(cs);
    (void)hipStreamDestroy(st);
}

}  // namespace

int main(int argc, char **argv)
{
    Options o;
    for (int a = 1; a < argc; ++a) {
        std::string k = argv[a];
        auto next = [&]() -> const char * { if (a + 1 >= argc) { fprintf(stderr, "missing value for %s\n", k.c_str()); exit(2); } return argv[++a]; };
        if (k == "--gpus") o.gpus = atoi(next());
        else if (k == "--col-shards") o.col_shards = atoi(next());
        else if (k == "--share-gpu") o.share_gpu = true;
        else if (k == "--transport") { std::string v = next(); o.transport = v == "rccl" ? MS_DIST_RCCL : (v == "host" ? MS_DIST_HOST : MS_DIST_AUTO); }
        else if (k == "--frames") o.frames = atoi(next());
        else if (k == "--batch") o.batch = atoi(next());
        else if (k == "--views") o.views = atoi(next());
        else if (k == "--size") sscanf(next(), "%dx%d", &o.w, &o.h);
        else if (k == "--out") sscanf(next(), "%dx%d", &o.out_w, &o.out_h);
        else if (k == "--hfov") o.hfov = atof(next());
        else if (k == "--bands") o.bands = atoi(next());
        else if (k == "--cpw") o.cpw = true;
        else if (k == "--recalib-every") { o.cpw = true; o.recalib_every = atoi(next()); }
        else if (k == "--mesh") sscanf(next(), "%dx%d", &o.mesh_rows, &o.mesh_cols);
        else if (k == "--no-checksum") o.checksum = false;
        else if (k == "--frame-sums") o.frame_sums = true;
        else if (k == "--tables-from-rank0") o.tables_from_rank0 = true;
        else if (k == "--rccl-lib") o.rccl_lib = next();
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fprintf(stderr, "stitch_dist: no HIP device (libmsstitch has no CPU fallback)\n"); return 3; }
    if (o.tables_from_rank0 && o.col_shards > 1) { fprintf(stderr, "stitch_dist: --tables-from-rank0 needs --col-shards 1 (a blob carries its context's column window)\n"); return 2; }
    if (o.gpus < 1 || o.gpus > MS_DIST_MAX_RANKS || o.col_shards < 1 || o.gpus % o.col_shards != 0 || o.batch < 1 || o.frames < 1) { fprintf(stderr, "stitch_dist: --gpus must be a multiple of --col-shards\n"); return 2; }
    if (!o.share_gpu && o.gpus > ndev) { fprintf(stderr, "stitch_dist: %d ranks but %d devices (use --share-gpu to put every rank on device 0 over the host transport)\n", o.gpus, ndev); return 2; }
    const long long batch_frames = (long long)(o.gpus / o.col_shards) * o.batch;
    if (o.recalib_every > 0 && o.recalib_every % batch_frames != 0) { fprintf(stderr, "stitch_dist: --recalib-every must be a multiple of groups x batch = %lld (meshes swap between ms_stitch calls)\n", batch_frames); return 2; }
    Shared sh;
    int transport = o.transport;
    if (o.share_gpu && o.gpus > 1 && transport == MS_DIST_AUTO) transport = MS_DIST_HOST;       // RCCL refuses two ranks on one device (an explicit --transport rccl is honoured: the loopback library of the tests does not)
    if (!o.rccl_lib.empty() && ms_dist_set_rccl_library(o.rccl_lib.c_str()) < 0) { fprintf(stderr, "stitch_dist: %s\n", ms_last_error()); return 1; }
    if (ms_dist_unique_id(transport, o.gpus, sh.id) < 0) { fprintf(stderr, "stitch_dist: %s\n", ms_last_error()); return 1; }
    std::vector<std::thread> ranks;
    for (int r = 0; r < o.gpus; ++r)
        ranks.emplace_back([&, r] {
            try { rank_main(o, r, sh); }
            catch (const std::exception &e) { std::lock_guard<std::mutex> lk(sh.mu); if (sh.failure.empty()) sh.failure = "rank " + std::to_string(r) + ": " + e.what(); }
        });
    for (auto &t : ranks) t.join();
    if (!sh.failure.empty()) { fprintf(stderr, "stitch_dist: %s\n", sh.failure.c_str()); return 1; }
    unsigned long long all = 1469598103934665603ull;
    for (unsigned long long s : sh.frame_sums) all = fnv(reinterpret_cast<const unsigned char *>(&s), sizeof(s), all);
    if (o.frame_sums) { for (size_t i = 0; i < sh.frame_sums.size(); ++i) fprintf(stderr, "frame %zu %016llx\n", i, sh.frame_sums[i]); }
    std::string devs = "[", pcis = "[", reads = "[";
    for (int r = 0; r < o.gpus; ++r) {
        devs += (r ? ", " : "") + std::to_string(sh.info.device[r]);
        pcis += std::string(r ? ", " : "") + "\"" + sh.info.pci_bus_id[r] + "\"";
        reads += (r ? ", " : "") + std::to_string(sh.views_read[r]);
    }
    devs += "]"; pcis += "]"; reads += "]";
    std::string times = "[";
    for (int r = 0; r < o.gpus; ++r) {
        char b[320];
        const Shared::RankTimes &t = sh.times[r];
        snprintf(b, sizeof(b), "%s{\"rank\": %d, \"wall_ms\": %.2f, \"stitch_gpu_ms\": %.2f, \"gather_stream_ms\": %.2f, \"mesh_exchange_host_ms\": %.2f, \"shard_exchange_host_ms\": %.2f, \"consume_host_ms\": %.2f}",
                 r ? ", " : "", r, t.wall_ms, t.stitch_gpu_ms, t.gather_stream_ms, t.mesh_exchange_host_ms, t.shard_exchange_host_ms, t.consume_host_ms);
        times += b;
    }
    times += "]";
    const long long frames_done = o.checksum ? (long long)sh.frame_sums.size() : o.frames;
    char rccl_path[1024] = "";      // the file the RCCL entry points were resolved from (host transport: never resolved, stays empty)
    if (sh.info.transport == MS_DIST_RCCL) (void)ms_dist_rccl_library_path(rccl_path, sizeof rccl_path);
    printf("{\"app\": \"stitch_dist\", \"gpus\": %d, \"col_shards\": %d, \"groups\": %d, \"share_gpu\": %s, \"views\": %d, \"src\": \"%dx%d\", \"out\": \"%dx%d\", \"bands\": %d, "
           "\"cpw\": %s, \"recalib_every\": %d, \"recalibrations_applied\": %d, \"batch\": %d, \"frames\": %lld, \"seconds\": %.4f, \"frames_per_s\": %.1f, "
           "\"dist\": {\"transport\": \"%s\", \"nranks\": %d, \"comm_nranks\": %d, \"rccl_version\": %d, \"librccl_path\": \"%s\", \"devices\": %s, \"pci_bus_ids\": %s}, "
           "\"views_read_per_rank\": %s, \"per_rank\": %s, \"i420_rows\": %d, \"first_frame_checksum\": \"%016llx\", \"last_frame_checksum\": \"%016llx\", \"checksum_all\": \"%016llx\"}\n",
           o.gpus, o.col_shards, o.gpus / o.col_shards, o.share_gpu ? "true" : "false", o.views, o.w, o.h, o.out_w, o.out_h, sh.bands,
           o.cpw ? "true" : "false", o.recalib_every, sh.recalibrations, o.batch, frames_done, sh.seconds, frames_done / sh.seconds,
           sh.info.transport == MS_DIST_RCCL ? "rccl" : "host", sh.info.nranks, sh.info.comm_nranks, sh.info.rccl_version, rccl_path, devs.c_str(), pcis.c_str(),
           reads.c_str(), times.c_str(), sh.i_rows, sh.frame_sums.empty() ? 0ull : sh.frame_sums.front(), sh.frame_sums.empty() ? 0ull : sh.frame_sums.back(), all);
    return 0;
}
