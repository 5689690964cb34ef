// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY: a loopback stand-in for librccl with RCCL's own prototypes (<rccl/rccl.h>), so that the RCCL branch of
// csrc/dist.cpp (rccl_attach, ncclSend / ncclRecv inside ncclGroupStart / End, ncclBroadcast, ncclAllGather) runs with N > 1 ranks on a box with ONE
// GPU -- the real library refuses two ranks on one device.  It proves call order, grouping, argument marshalling and stream / event ordering of the
// callers; it says NOTHING about RCCL itself or about xGMI (tests/test_ms_dist_gpu.py keeps calling the N > 1 RCCL transport "unmeasured").
//
// Semantics kept from NCCL: every call only ENQUEUES.  A transfer starts when the work enqueued before it on its stream has finished (a hipEvent the
// proxy thread waits for) and the stream continues behind it only when the transfer has completed (a one-lane kernel on the stream spins on a word of
// pinned host memory that the proxy thread advances) -- so a caller that forgets a stream / event dependency reads stale bytes here as it would with
// the real library, and a caller that blocks the host where RCCL would not shows up as a deadlock.  The bytes themselves travel through a POSIX
// shared-memory mailbox (ranks = threads or processes), one single-slot channel per ordered rank pair, matched in posting order per channel.
// Loaded through ms_dist_set_rccl_library(path); never on any default search path, never linked into the product.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr size_t CHUNK = 1u << 20;
constexpr double TIMEOUT_S = 60.0;
constexpr unsigned MAGIC = 0x46524343u;      // "FRCC"
constexpr int FAKE_VERSION = 29999;          // major 2 (dist.cpp checks it), and recognisable as not a real release

struct alignas(64) Chan {
    std::atomic<unsigned long long> head, tail;
    unsigned long long len;
    unsigned char pad[64 - 24];
    unsigned char data[CHUNK];
};
struct alignas(64) Header {
    std::atomic<unsigned> magic, attached, detached;
    unsigned nranks;
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Backoff {
    int n = 0;
    double t0 = now_s();
    bool wait()
    {
        if (++n < 200) return true;
        if (n < 2000) { sched_yield(); return true; }
        usleep(50);
        return (n & 1023) != 0 || now_s() - t0 < TIMEOUT_S;
    }
};

enum Kind { SEND, RECV, COPY };
struct Op {
    Kind kind;
    unsigned char *buf;                   // SEND: source, RECV / COPY: destination
    const unsigned char *src = nullptr;   // COPY only
    size_t bytes;
    int peer;
    size_t done = 0;
    bool finished = false;
};
struct Batch {
    std::vector<Op> ops;
    std::vector<hipEvent_t> ready;        // recorded on every stream of the batch before the wait kernel: the transfer starts behind them
    unsigned long long seq;
};

__global__ void k_fake_rccl_wait(const unsigned long long *flag, unsigned long long seq, unsigned *err)
{
    const unsigned long long t0 = wall_clock64();                      // 100 MHz
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        __builtin_amdgcn_s_sleep(64);
        if (wall_clock64() - t0 > 100000000ull * (unsigned long long)TIMEOUT_S) { __hip_atomic_store(err, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
}

size_t dtype_size(ncclDataType_t t)
{
    switch ((int)t) {
    case 0: case 1: case 10: case 11: return 1;
    case 6: case 9: return 2;
    case 2: case 3: case 7: return 4;
    case 4: case 5: case 8: return 8;
    default: return 0;
    }
}

}  // namespace

struct ncclComm {
    int rank = 0, nranks = 1, device = 0;
    Header *shm = nullptr;
    size_t shm_len = 0;
    bool registered = false;
    hipStream_t priv = nullptr;
    unsigned long long *flag = nullptr;   // pinned host: sequence number of the last completed batch
    unsigned *err = nullptr;              // pinned host: set by a wait kernel that timed out
    unsigned long long submitted = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Batch> queue;
    bool stop = false;
    std::atomic<int> failed{0};
    std::thread worker;
    Chan *chan(int src, int dst) const { return reinterpret_cast<Chan *>(reinterpret_cast<unsigned char *>(shm) + sizeof(Header)) + ((size_t)src * nranks + dst); }
};

namespace {

thread_local int g_depth = 0;
struct Pending { ncclComm *comm; hipStream_t stream; Op op; };
thread_local std::vector<Pending> g_pending;

bool copy_sync(ncclComm *c, void *dst, const void *src, size_t n, hipMemcpyKind kind)
{
    if (n == 0) return true;
    return hipMemcpyAsync(dst, src, n, kind, c->priv) == hipSuccess && hipStreamSynchronize(c->priv) == hipSuccess;
}

// one batch = the operations of one group (or one ungrouped call) on this communicator: all of them progress together, per channel in posting order
bool run_batch(ncclComm *c, Batch &b)
{
    for (hipEvent_t e : b.ready) if (hipEventSynchronize(e) != hipSuccess) return false;      // (destroyed by worker_main)
    std::vector<Op> &ops = b.ops;
    for (Op &o : ops) if (o.kind == COPY) { if (!copy_sync(c, o.buf, o.src, o.bytes, hipMemcpyDeviceToDevice)) return false; o.finished = true; }
    for (size_t i = 0; i < ops.size(); ++i) {                                  // a send to oneself pairs with the receive from oneself of the same batch
        if (ops[i].kind != SEND || ops[i].peer != c->rank || ops[i].finished) continue;
        for (size_t j = 0; j < ops.size(); ++j)
            if (ops[j].kind == RECV && ops[j].peer == c->rank && !ops[j].finished && ops[j].bytes == ops[i].bytes) {
                if (!copy_sync(c, ops[j].buf, ops[i].buf, ops[i].bytes, hipMemcpyDeviceToDevice)) return false;
                ops[i].finished = ops[j].finished = true;
                break;
            }
        if (!ops[i].finished) return false;
    }
    Backoff bo;
    for (;;) {
        bool all = true, any = false;
        for (size_t i = 0; i < ops.size(); ++i) {
            Op &o = ops[i];
            if (o.finished) continue;
            bool blocked = false;
            for (size_t j = 0; j < i && !blocked; ++j) blocked = !ops[j].finished && ops[j].peer == o.peer && ops[j].kind == o.kind;
            if (blocked) { all = false; continue; }
            Chan *ch = o.kind == SEND ? c->chan(c->rank, o.peer) : c->chan(o.peer, c->rank);
            const unsigned long long h = ch->head.load(std::memory_order_acquire), t = ch->tail.load(std::memory_order_acquire);
            if (o.kind == SEND) {
                if (h != t) { all = false; continue; }
                const size_t n = std::min(CHUNK, o.bytes - o.done);
                if (!copy_sync(c, ch->data, o.buf + o.done, n, hipMemcpyDeviceToHost)) return false;
                ch->len = n;
                ch->head.store(h + 1, std::memory_order_release);
                o.done += n;
            } else {
                if (h == t) { all = false; continue; }
                const size_t n = (size_t)ch->len;
                if (n > o.bytes - o.done) { fprintf(stderr, "fake_rccl: rank %d receives %zu bytes from %d, %zu posted\n", c->rank, n, o.peer, o.bytes - o.done); return false; }
                if (!copy_sync(c, o.buf + o.done, ch->data, n, hipMemcpyHostToDevice)) return false;
                ch->tail.store(t + 1, std::memory_order_release);
                o.done += n;
            }
            any = true;
            o.finished = o.done == o.bytes;
            all &= o.finished;
        }
        if (all) return true;
        if (any) { bo = Backoff(); continue; }
        if (!bo.wait()) { fprintf(stderr, "fake_rccl: rank %d timed out waiting for a peer\n", c->rank); return false; }
    }
}

void worker_main(ncclComm *c)
{
    (void)hipSetDevice(c->device);
    for (;;) {
        Batch b;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv.wait(lk, [&] { return c->stop || !c->queue.empty(); });
            if (c->queue.empty()) return;
            b = std::move(c->queue.front());
            c->queue.pop_front();
        }
        if (!c->failed.load() && !run_batch(c, b)) c->failed.store(1);
        for (hipEvent_t e : b.ready) (void)hipEventDestroy(e);
        __atomic_store_n(c->flag, b.seq, __ATOMIC_RELEASE);           // (also after a failure: the streams must not hang; the next call reports it)
        c->cv.notify_all();
    }
}

// the calling thread's part: events on the streams, the batch to the proxy thread, one wait kernel per stream
ncclResult_t submit(ncclComm *c, std::vector<Op> ops, const std::vector<hipStream_t> &streams)
{
    if (c->failed.load() || __atomic_load_n(c->err, __ATOMIC_ACQUIRE)) return ncclSystemError;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev != c->device) (void)hipSetDevice(c->device);
    Batch b;
    b.ops = std::move(ops);
    std::vector<hipStream_t> uniq;
    for (hipStream_t s : streams) { bool seen = false; for (hipStream_t u : uniq) seen |= u == s; if (!seen) uniq.push_back(s); }
    for (hipStream_t s : uniq) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess || hipEventRecord(e, s) != hipSuccess) return ncclUnhandledCudaError;
        b.ready.push_back(e);
    }
    {
        std::lock_guard<std::mutex> lk(c->mu);
        b.seq = ++c->submitted;
        c->queue.push_back(std::move(b));
    }
    const unsigned long long seq = c->submitted;
    c->cv.notify_all();
    for (hipStream_t s : uniq) hipLaunchKernelGGL(k_fake_rccl_wait, dim3(1), dim3(1), 0, s, c->flag, seq, c->err);
    if (dev != c->device) (void)hipSetDevice(dev);
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t post(ncclComm *c, hipStream_t s, const std::vector<Op> &ops)
{
    if (!c) return ncclInvalidArgument;
    if (g_depth > 0) { for (const Op &o : ops) g_pending.push_back(Pending{c, s, o}); return ncclSuccess; }
    return submit(c, ops, std::vector<hipStream_t>(1, s));
}

void shm_name(const ncclUniqueId &id, char out[48])
{
    unsigned long long a = 0, b = 0;
    memcpy(&a, id.internal, 8); memcpy(&b, id.internal + 8, 8);
    snprintf(out, 48, "/fakerccl_%016llx%016llx", a, b);
}

}  // namespace

extern "C" {

#define FR_API __attribute__((visibility("default")))

FR_API ncclResult_t ncclGetVersion(int *v) { if (!v) return ncclInvalidArgument; *v = FAKE_VERSION; return ncclSuccess; }
FR_API const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake_rccl: HIP call failed";
    case ncclSystemError: return "fake_rccl: a peer timed out or an earlier transfer failed";
    case ncclInvalidArgument: return "fake_rccl: invalid argument";
    default: return "fake_rccl: error";
    }
}
FR_API ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    FILE *f = fopen("/dev/urandom", "rb");
    const size_t got = f ? fread(id->internal, 1, sizeof(id->internal), f) : 0;
    if (f) fclose(f);
    return got == sizeof(id->internal) ? ncclSuccess : ncclSystemError;
}

FR_API ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > 16 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (const char *h = getenv("FAKE_RCCL_HANG_S")) sleep((unsigned)atoi(h));
    ncclComm *c = new ncclComm();
    c->rank = rank; c->nranks = nranks;
    (void)hipGetDevice(&c->device);
    char name[48];
    shm_name(id, name);
    const size_t len = sizeof(Header) + sizeof(Chan) * (size_t)nranks * nranks;
    int fd = -1;
    if (rank == 0) {
        shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)len) != 0) { if (fd >= 0) { close(fd); shm_unlink(name); } delete c; return ncclSystemError; }
    } else {
        Backoff bo;
        for (;;) {
            fd = shm_open(name, O_RDWR, 0600);
            struct stat sb;
            if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size == len) break;
            if (fd >= 0) { close(fd); fd = -1; }
            if (!bo.wait()) { delete c; return ncclSystemError; }
        }
    }
    void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { if (rank == 0) shm_unlink(name); delete c; return ncclSystemError; }
    c->shm = static_cast<Header *>(p); c->shm_len = len;
    if (rank == 0) { c->shm->nranks = (unsigned)nranks; c->shm->magic.store(MAGIC, std::memory_order_release); }
    Backoff bo;
    while (c->shm->magic.load(std::memory_order_acquire) != MAGIC) if (!bo.wait()) { munmap(p, len); delete c; return ncclSystemError; }
    c->shm->attached.fetch_add(1, std::memory_order_acq_rel);
    while (c->shm->attached.load(std::memory_order_acquire) < (unsigned)nranks) if (!bo.wait()) { if (rank == 0) shm_unlink(name); munmap(p, len); delete c; return ncclSystemError; }
    if (rank == 0) shm_unlink(name);                                  // every rank has mapped it: nothing stays in /dev/shm
    c->registered = hipHostRegister(p, len, hipHostRegisterDefault) == hipSuccess;       // (pageable staging works too, only slower)
    if (!c->registered) (void)hipGetLastError();
    bool ok = hipStreamCreateWithFlags(&c->priv, hipStreamNonBlocking) == hipSuccess;    // non-blocking: a wait kernel on the NULL stream must not hold the proxy's copies
    ok = ok && hipHostMalloc((void **)&c->flag, 64, hipHostMallocDefault) == hipSuccess;
    if (!ok) { delete c; return ncclUnhandledCudaError; }
    c->flag[0] = 0;
    c->err = reinterpret_cast<unsigned *>(c->flag + 4);
    *c->err = 0;
    c->worker = std::thread(worker_main, c);
    *out = c;
    return ncclSuccess;
}

FR_API ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    {
        std::unique_lock<std::mutex> lk(c->mu);
        c->cv.wait(lk, [&] { return c->queue.empty() && __atomic_load_n(c->flag, __ATOMIC_ACQUIRE) >= c->submitted; });
        c->stop = true;
    }
    c->cv.notify_all();
    if (c->worker.joinable()) c->worker.join();
    (void)hipDeviceSynchronize();                                      // the wait kernels of the last batches have left the streams
    if (c->priv) (void)hipStreamDestroy(c->priv);
    if (c->flag) (void)hipHostFree(c->flag);
    if (c->shm) { if (c->registered) (void)hipHostUnregister(c->shm); munmap(c->shm, c->shm_len); }
    delete c;
    return ncclSuccess;
}

// Two fault injections for the bench's fail-loudly paths (tests/test_bench_gpu.py), read from the environment of THIS test library only:
//   FAKE_RCCL_COUNT_DELTA=d   ncclCommCount reports nranks + d (a communicator that did not form over all ranks)
//   FAKE_RCCL_HANG_S=s        ncclCommInitRank sleeps s seconds before it does anything (a bring-up that hangs: the bench's watchdog must end the run)
FR_API ncclResult_t ncclCommCount(const ncclComm_t c, int *n)
{
    if (!c || !n) return ncclInvalidArgument;
    const char *d = getenv("FAKE_RCCL_COUNT_DELTA");
    *n = c->nranks + (d ? atoi(d) : 0);
    return ncclSuccess;
}

FR_API ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    if (!c || peer < 0 || peer >= c->nranks || !dtype_size(t) || (!buf && count)) return ncclInvalidArgument;
    if (peer == c->rank && g_depth == 0) return ncclInvalidUsage;
    return post(c, s, {Op{SEND, static_cast<unsigned char *>(const_cast<void *>(buf)), nullptr, count * dtype_size(t), peer}});
}
FR_API ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    if (!c || peer < 0 || peer >= c->nranks || !dtype_size(t) || (!buf && count)) return ncclInvalidArgument;
    if (peer == c->rank && g_depth == 0) return ncclInvalidUsage;
    return post(c, s, {Op{RECV, static_cast<unsigned char *>(buf), nullptr, count * dtype_size(t), peer}});
}
FR_API ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t s)
{
    if (!c || root < 0 || root >= c->nranks || !dtype_size(t) || !recv) return ncclInvalidArgument;
    const size_t n = count * dtype_size(t);
    std::vector<Op> ops;
    if (c->rank == root) {
        if (!send) return ncclInvalidArgument;
        if (send != recv) ops.push_back(Op{COPY, static_cast<unsigned char *>(recv), static_cast<const unsigned char *>(send), n, root});
        for (int r = 0; r < c->nranks; ++r) if (r != root) ops.push_back(Op{SEND, static_cast<unsigned char *>(const_cast<void *>(send)), nullptr, n, r});
    } else ops.push_back(Op{RECV, static_cast<unsigned char *>(recv), nullptr, n, root});
    return post(c, s, ops);
}
FR_API ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s)
{
    if (!c || !dtype_size(t) || !send || !recv) return ncclInvalidArgument;
    const size_t n = count * dtype_size(t);
    unsigned char *out = static_cast<unsigned char *>(recv);
    std::vector<Op> ops;
    if (out + (size_t)c->rank * n != send) ops.push_back(Op{COPY, out + (size_t)c->rank * n, static_cast<const unsigned char *>(send), n, c->rank});
    for (int r = 0; r < c->nranks; ++r) if (r != c->rank) ops.push_back(Op{SEND, static_cast<unsigned char *>(const_cast<void *>(send)), nullptr, n, r});
    for (int r = 0; r < c->nranks; ++r) if (r != c->rank) ops.push_back(Op{RECV, out + (size_t)r * n, nullptr, n, r});
    return post(c, s, ops);
}

FR_API ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
FR_API ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Pending> pend;
    pend.swap(g_pending);
    ncclResult_t res = ncclSuccess;
    std::vector<bool> taken(pend.size(), false);
    for (size_t i = 0; i < pend.size(); ++i) {                         // one batch per communicator, operations in posting order
        if (taken[i]) continue;
        std::vector<Op> ops;
        std::vector<hipStream_t> streams;
        for (size_t j = i; j < pend.size(); ++j)
            if (!taken[j] && pend[j].comm == pend[i].comm) { taken[j] = true; ops.push_back(pend[j].op); streams.push_back(pend[j].stream); }
        const ncclResult_t r = submit(pend[i].comm, std::move(ops), streams);
        if (r != ncclSuccess) res = r;
    }
    return res;
}

}  // extern "C"
