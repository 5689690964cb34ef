// dist_tsan_check.cpp -- ThreadSanitizer run of ms_dist's host transport, one thread per rank (how stitch_dist --share-gpu and stitch_app --gpus N
// drive it).  Test infrastructure: compiled by tests/test_ms_dist.py::test_host_transport_under_tsan together with csrc/dist.cpp
//   hipcc -fsanitize=thread -g -O1 tests/dist_tsan_check.cpp video-stitcher_amd/csrc/dist.cpp -L video-stitcher_amd -lmsstitch ...
// (dist.cpp instrumented, the rest of the library as built).  Host-memory messages only: runs without a GPU.
// What it exercises per iteration, on every rank: a group with TWO sends and TWO receives per peer (the per-channel posting order of round 4), a
// broadcast from a rotating root, the frame gather into a rotating sink, a barrier, and the mesh exchange with an update every third iteration.
// Exit code 0 = every payload arrived intact (TSan itself makes the process exit non-zero on a report: TSAN_OPTIONS=halt_on_error=1).
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "../include/ms_dist.h"

namespace ms {      // the two error helpers dist.cpp takes from the library's api.cpp (hidden there): local versions for this binary
static thread_local char g_err[512];
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
int fail(int code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return code; }
}

static std::atomic<int> g_bad{0};
#define CHECK(expr) do { const int r_ = (expr); if (r_ != 0) { fprintf(stderr, "rank %d: %s -> %d (%s)\n", rank, #expr, r_, ms::g_err); g_bad++; return; } } while (0)

static uint32_t word(int from, int to, int iter, int slot, size_t i) { return (uint32_t)(from * 1000003u + to * 10007u + iter * 101u + slot * 7u) ^ (uint32_t)(i * 2654435761u); }

static void rank_main(int rank, int world, const void *id, int iters)
{
    ms_dist *d = nullptr;
    CHECK(ms_dist_create(&d, rank, world, id, -1));
    const size_t n = 3000;      // words per message (12 kB: several mailbox chunks)
    std::vector<std::vector<uint32_t>> tx(2 * world, std::vector<uint32_t>(n)), rx(2 * world, std::vector<uint32_t>(n));
    std::vector<uint32_t> bc(n), slab(n);
    std::vector<std::vector<uint32_t>> got(world, std::vector<uint32_t>(n));
    const int nv = 2, rows = 5, cols = 4;
    std::vector<float> mx(nv * rows * cols), my(nv * rows * cols), ox(nv * rows * cols), oy(nv * rows * cols);
    for (int it = 0; it < iters; ++it) {
        // two messages per peer and direction in one group, posted in a different order on the two sides
        CHECK(ms_dist_group_begin(d));
        for (int p = 0; p < world; ++p) {
            if (p == rank) continue;
            for (int s = 0; s < 2; ++s) {
                for (size_t i = 0; i < n; ++i) tx[2 * p + s][i] = word(rank, p, it, s, i);
                CHECK(ms_dist_send(d, tx[2 * p + s].data(), n * 4, p, MS_DIST_MEM_HOST, nullptr));
            }
        }
        for (int p = world - 1; p >= 0; --p) {
            if (p == rank) continue;
            for (int s = 0; s < 2; ++s) CHECK(ms_dist_recv(d, rx[2 * p + s].data(), n * 4, p, MS_DIST_MEM_HOST, nullptr));
        }
        CHECK(ms_dist_group_end(d));
        for (int p = 0; p < world; ++p)
            for (int s = 0; s < 2 && p != rank; ++s)
                for (size_t i = 0; i < n; ++i)
                    if (rx[2 * p + s][i] != word(p, rank, it, s, i)) { fprintf(stderr, "rank %d: message %d of peer %d differs at word %zu (iteration %d)\n", rank, s, p, i, it); g_bad++; return; }
        // broadcast from a rotating root
        const int root = it % world;
        for (size_t i = 0; i < n; ++i) bc[i] = rank == root ? word(root, 99, it, 3, i) : 0u;
        CHECK(ms_dist_broadcast(d, bc.data(), n * 4, root, MS_DIST_MEM_HOST, nullptr));
        for (size_t i = 0; i < n; ++i) if (bc[i] != word(root, 99, it, 3, i)) { fprintf(stderr, "rank %d: broadcast differs (iteration %d)\n", rank, it); g_bad++; return; }
        CHECK(ms_dist_barrier(d, nullptr));
        // mesh exchange: an update from the root every third iteration
        ms_dist_mesh_update upd{}, out{};
        out.mesh_x = ox.data(); out.mesh_y = oy.data();
        const bool has = it % 3 == 0;
        if (rank == 0 && has) {
            for (size_t i = 0; i < mx.size(); ++i) { mx[i] = (float)(it * 1000 + (int)i); my[i] = -(float)(it * 1000 + (int)i); }
            upd.swap_frame = 16LL * it; upd.version = it; upd.n_views = nv; upd.rows = rows; upd.cols = cols; upd.mesh_x = mx.data(); upd.mesh_y = my.data();
        }
        int have = -1;
        CHECK(ms_dist_mesh_exchange(d, 0, rank == 0 && has ? &upd : nullptr, &out, ox.size(), &have, nullptr));
        if (have != (has ? 1 : 0)) { fprintf(stderr, "rank %d: mesh exchange have = %d (iteration %d)\n", rank, have, it); g_bad++; return; }
        if (has && (out.version != it || out.swap_frame != 16LL * it || ox[7] != (float)(it * 1000 + 7) || oy[11] != -(float)(it * 1000 + 11))) {
            fprintf(stderr, "rank %d: mesh update differs (iteration %d)\n", rank, it); g_bad++; return;
        }
    }
    (void)slab; (void)got;
    CHECK(ms_dist_barrier(d, nullptr));
    ms_dist_destroy(d);
}

int main(int argc, char **argv)
{
    const int world = argc > 1 ? atoi(argv[1]) : 3, iters = argc > 2 ? atoi(argv[2]) : 40;
    unsigned char id[MS_DIST_ID_BYTES];
    if (ms_dist_unique_id(MS_DIST_HOST, world, id) != 0) { fprintf(stderr, "ms_dist_unique_id failed: %s\n", ms::g_err); return 2; }
    std::vector<std::thread> ts;
    for (int r = 0; r < world; ++r) ts.emplace_back(rank_main, r, world, (const void *)id, iters);
    for (auto &t : ts) t.join();
    if (g_bad.load()) { fprintf(stderr, "FAILED\n"); return 1; }
    printf("ok: %d ranks x %d iterations\n", world, iters);
    return 0;
}
