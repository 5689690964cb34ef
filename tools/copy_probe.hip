// copy_probe.hip -- what does a streaming copy reach on this MI355X?  (the ceiling every "runs at copy rate"
// argument in DESIGN.md section 5 is measured against must itself be a tuned stream, not a one-load-in-flight loop.)
//   hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o ab/copy_probe && ab/copy_probe [MiB]
// Variants: loads in flight per lane (U), workgroups per CU, workgroup size, non-temporal loads / stores, read-only, write-only.
// Prints one line per variant: bytes moved / best-of-5 hipEvent time.  Copy lines count read + written bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NTL, bool NTS>
__global__ void __launch_bounds__(1024) k_copy(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16)
{
    // every lane keeps U independent 16-byte loads in flight; a workgroup walks the buffer in chunks of U * blockDim.x vectors so that each
    // wave-instruction is one contiguous 1 KiB segment
    const size_t chunk = (size_t)U * blockDim.x;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n16; base += (size_t)gridDim.x * chunk) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
            if (i < n16) v[u] = NTL ? __builtin_nontemporal_load(src + i) : src[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
            if (i < n16) { if (NTS) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; }
        }
    }
}

template <int U, bool NTL>
__global__ void __launch_bounds__(1024) k_read(const u32x4 *__restrict__ src, unsigned *__restrict__ sink, size_t n16)
{
    const size_t chunk = (size_t)U * blockDim.x;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (size_t base = (size_t)blockIdx.x * chunk; base < n16; base += (size_t)gridDim.x * chunk) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
            v[u] = i < n16 ? (NTL ? __builtin_nontemporal_load(src + i) : src[i]) : acc;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1u;      // (never, in practice: keeps the loads alive)
}

template <int U, bool NTS>
__global__ void __launch_bounds__(1024) k_write(u32x4 *__restrict__ dst, size_t n16)
{
    const size_t chunk = (size_t)U * blockDim.x;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (size_t base = (size_t)blockIdx.x * chunk; base < n16; base += (size_t)gridDim.x * chunk)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
            if (i < n16) { if (NTS) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
        }
}

static hipEvent_t e0, e1;
template <class F>
static double best_ms(F launch)
{
    double best = 1e30;
    for (int it = 0; it < 7; ++it) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) best = std::min(best, (double)ms);
    }
    CK(hipGetLastError());
    return best;
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)atoll(argv[1]) : 1024;
    const size_t bytes = mib << 20, n16 = bytes / 16;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs, buffer %zu MiB (read + write streams of that size)\n", prop.name, cus, mib);
    u32x4 *a, *b;
    unsigned *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 0x5a, bytes)); CK(hipMemset(b, 0, bytes));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto report = [&](const char *what, int U, int wg, int per_cu, const char *pol, double ms, double moved) {
        printf("%-6s U=%d wg=%-4d blocks/CU=%-3d %-9s %8.1f us  %6.3f TB/s\n", what, U, wg, per_cu, pol, ms * 1e3, moved / (ms * 1e-3) / 1e12);
        fflush(stdout);
    };
    {   // the r02 calibration kernel's shape: 2048 x 256 lanes, one load in flight
        const double ms = best_ms([&] { k_copy<1, false, false><<<2048, 256>>>(a, b, n16); });
        report("copy", 1, 256, 2048 / cus, "r02-shape", ms, 2.0 * bytes);
    }
    const int wgs[] = {256, 512, 1024};
    const int percu[] = {1, 2, 4, 8, 16};
    for (int wg : wgs)
        for (int pc : percu) {
            if (wg * pc > 2048 * 2) continue;
            const int grid = cus * pc;
#define RUN_COPY(U)                                                                                                           \
            report("copy", U, wg, pc, "plain", best_ms([&] { k_copy<U, false, false><<<grid, wg>>>(a, b, n16); }), 2.0 * bytes); \
            report("copy", U, wg, pc, "nt-store", best_ms([&] { k_copy<U, false, true><<<grid, wg>>>(a, b, n16); }), 2.0 * bytes); \
            report("copy", U, wg, pc, "nt-both", best_ms([&] { k_copy<U, true, true><<<grid, wg>>>(a, b, n16); }), 2.0 * bytes);
            RUN_COPY(2) RUN_COPY(4) RUN_COPY(8)
        }
    for (int pc : {2, 4, 8, 16}) {
        const int grid = cus * pc;
        report("read", 4, 256, pc, "plain", best_ms([&] { k_read<4, false><<<grid, 256>>>(a, sink, n16); }), 1.0 * bytes);
        report("read", 8, 256, pc, "plain", best_ms([&] { k_read<8, false><<<grid, 256>>>(a, sink, n16); }), 1.0 * bytes);
        report("read", 8, 256, pc, "nt", best_ms([&] { k_read<8, true><<<grid, 256>>>(a, sink, n16); }), 1.0 * bytes);
        report("write", 4, 256, pc, "plain", best_ms([&] { k_write<4, false><<<grid, 256>>>(b, n16); }), 1.0 * bytes);
        report("write", 4, 256, pc, "nt", best_ms([&] { k_write<4, true><<<grid, 256>>>(b, n16); }), 1.0 * bytes);
    }
    {
        const double ms = best_ms([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
        report("hipMemcpyDtoD", 0, 0, 0, "runtime", ms, 2.0 * bytes);
    }
    return 0;
}
