// matcher.hip -- brute-force Hamming 2-nearest-neighbour matching on the device (gfx950): what featurefinder::matchFeatures asks of
// DescriptorMatcher::create("BruteForce-Hamming")->knnMatch(f1.descriptors, f2.descriptors, matches, 2)  (360_stitcher/featurefinder.cpp:50-61;
// BFMatcher::knnMatchImpl, features2d/src/matchers.cpp:815-880 -> cv::batchDistance, core/src/stat.cpp:3946-4008).
//
// Integer work, exact: the K = 2 insertion of BatchDistInvoker keeps, for every query row, the two train rows that come first in
// (distance, train index) order (`d < dist[K-1]` is strict and equal distances are never moved), and that is what each wave computes:
// one wave per query, lanes stride over the train rows keeping a private best pair, then a 6-step butterfly merges the 64 pairs.
#include <climits>
#include <vector>
#include "common.hpp"

namespace ms {
namespace {

struct Best2 { int d0, j0, d1, j1; };

__device__ __forceinline__ bool before(int da, int ja, int db, int jb) { return da < db || (da == db && ja < jb); }

__device__ __forceinline__ void insert(Best2 &b, int d, int j)
{
    if (before(d, j, b.d0, b.j0)) { b.d1 = b.d0; b.j1 = b.j0; b.d0 = d; b.j0 = j; }
    else if (before(d, j, b.d1, b.j1)) { b.d1 = d; b.j1 = j; }
}

template <int WORDS>
__global__ void __launch_bounds__(256) k_hamming_knn2(const uint8_t *__restrict__ query, size_t qstep, int nq, const uint8_t *__restrict__ train, size_t tstep, int nt,
                                                      int words, int4 *__restrict__ out)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    const int W = WORDS ? WORDS : words;
    unsigned qw[WORDS ? WORDS : 16];
    const unsigned *qp = (const unsigned *)(query + (size_t)q * qstep);
#pragma unroll
    for (int k = 0; k < (WORDS ? WORDS : 16); ++k) qw[k] = k < W ? qp[k] : 0u;
    Best2 b{INT_MAX, INT_MAX, INT_MAX, INT_MAX};          // index INT_MAX = "none": sorts after every real row at distance INT_MAX
    for (int j = lane; j < nt; j += 64) {
        const unsigned *tp = (const unsigned *)(train + (size_t)j * tstep);
        int d = 0;
#pragma unroll
        for (int k = 0; k < (WORDS ? WORDS : 16); ++k)
            if (k < W) d += __popc(qw[k] ^ tp[k]);
        insert(b, d, j);
    }
    for (int off = 32; off; off >>= 1) {
        const int d0 = __shfl_xor(b.d0, off), j0 = __shfl_xor(b.j0, off), d1 = __shfl_xor(b.d1, off), j1 = __shfl_xor(b.j1, off);
        insert(b, d0, j0);
        insert(b, d1, j1);
    }
    if (lane == 0) out[q] = make_int4(b.j0 == INT_MAX ? -1 : b.j0, b.d0, b.j1 == INT_MAX ? -1 : b.j1, b.d1);
}

}  // namespace
}  // namespace ms

using namespace ms;

extern "C" int ms_knn_match_hamming2(const ms_image *query, const ms_image *train, int *train_idx_host, int *distance_host, ms_stream stream)
{
    if (int e = require_device()) return e;
    MS_CHECK(query && train && train_idx_host && distance_host, "ms_knn_match_hamming2: null argument");
    MS_CHECK(query->type == MS_8UC1 && train->type == MS_8UC1 && query->cols == train->cols, "ms_knn_match_hamming2: descriptors must be 8UC1 rows of equal length");
    MS_CHECK(query->cols > 0 && query->cols % 4 == 0 && query->cols <= 64, "ms_knn_match_hamming2: descriptor length %d (need a multiple of 4 up to 64 bytes)", query->cols);
    MS_CHECK(query->rows >= 0 && train->rows >= 0 && query->step % 4 == 0 && train->step % 4 == 0, "ms_knn_match_hamming2: rows must be 4-byte aligned");
    const int nq = query->rows, nt = train->rows;
    if (nq == 0) return MS_OK;
    MS_CHECK(query->data && (nt == 0 || train->data), "ms_knn_match_hamming2: null descriptors");
    hipStream_t st = as_stream(stream);
    int4 *out = (int4 *)device_scratch().get((size_t)nq * sizeof(int4));      // per-thread grow-only block: see DeviceScratch (common.hpp)
    if (!out) return fail(MS_ERR_NOMEM, "ms_knn_match_hamming2: cannot allocate device scratch");
    const int words = query->cols / 4;
    if (words == 8)          // ORB / BRIEF-32
        k_hamming_knn2<8><<<div_up(nq, 4), 256, 0, st>>>((const uint8_t *)query->data, query->step, nq, (const uint8_t *)train->data, train->step, nt, words, out);
    else
        k_hamming_knn2<0><<<div_up(nq, 4), 256, 0, st>>>((const uint8_t *)query->data, query->step, nq, (const uint8_t *)train->data, train->step, nt, words, out);
    hipError_t le = hipGetLastError();
    int4 *h = (int4 *)pinned_scratch().get((size_t)nq * sizeof(int4));
    hipError_t ce = hipSuccess;
    if (le == hipSuccess && h) {
        ce = hipMemcpyAsync(h, out, (size_t)nq * sizeof(int4), hipMemcpyDeviceToHost, st);
        if (ce == hipSuccess) ce = hipStreamSynchronize(st);
    }
    MS_HIP(le);
    if (!h) return fail(MS_ERR_NOMEM, "ms_knn_match_hamming2: cannot allocate pinned staging memory");
    MS_HIP(ce);
    for (int q = 0; q < nq; ++q) {        // batchDistance clamps K to the number of train rows: missing neighbours are index -1, distance INT_MAX
        train_idx_host[2 * q] = h[q].x; distance_host[2 * q] = h[q].y;
        train_idx_host[2 * q + 1] = h[q].z; distance_host[2 * q + 1] = h[q].w;
    }
    return MS_OK;
}
