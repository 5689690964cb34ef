// dist.cpp -- ms_dist: one rank per MI355X around the single-GPU compositor (include/ms_dist.h).
//
// RCCL transport: librccl.so.1 is dlopen'ed on first use (libmsstitch.so has no link-time dependency on it: a single-GPU caller never pays for
// it), one communicator per ms_dist, device buffers, stream-ordered calls, ncclGroupStart / End around concurrent sends and receives.
// xGMI is point-to-point (7 links x ~153 GB/s per GPU): the frame-parallel gather is G - 1 independent ncclSend -> one sink, each over its own
// link; nothing here is a ring, and nothing is on the per-frame compute path.
//
// HOST transport: ranks that share a device (RCCL refuses that) or a box without a second GPU: a POSIX shared-memory mailbox, one single-slot
// channel per ordered rank pair, blocking, staged through the host.  Works between threads and between processes; with MS_DIST_MEM_HOST it
// needs no device at all (tests/test_ms_dist.py runs the protocol on CPU).  A group advances all its operations round-robin, so two ranks
// that send to each other cannot deadlock on the single slot.
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "common.hpp"
#include "../../include/ms_dist.h"

#ifndef MS_ERR_COMM
#error "ms_stitch.h must define MS_ERR_COMM"
#endif

namespace ms {
namespace {

// ---------------------------------------------------------------------------------------------- RCCL, resolved at run time
// (the handful of declarations used, with RCCL's own ABI: rccl.h, ROCm 7.2)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2 };
struct Rccl {
    void *lib = nullptr;
    int (*GetVersion)(int *) = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(const ncclComm_t, int *) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
    const char *why = "";
};
std::mutex g_rccl_mu;
std::string g_rccl_path;            // ms_dist_set_rccl_library: an explicit file instead of the search below
bool g_rccl_resolved = false;
Rccl &rccl()
{
    static Rccl R = [] {
        Rccl r;
        std::string path;
        { std::lock_guard<std::mutex> lk(g_rccl_mu); path = g_rccl_path; g_rccl_resolved = true; }
        if (!path.empty()) {
            r.lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!r.lib) { r.why = "the library given to ms_dist_set_rccl_library could not be loaded"; return r; }
        } else {
            // a copy that is already in the process first (PyTorch ships its own librccl with the same soname: one RCCL per process, not two)
            const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char *n : names) if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
            if (!r.lib) for (const char *n : names) if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
            if (!r.lib) { r.why = "librccl.so.1 not found"; return r; }
        }
#define MS_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, sym)); if (!r.field) { r.why = "librccl lacks " sym; return r; }
        MS_SYM(GetVersion, "ncclGetVersion") MS_SYM(GetUniqueId, "ncclGetUniqueId") MS_SYM(CommInitRank, "ncclCommInitRank")
        MS_SYM(CommDestroy, "ncclCommDestroy") MS_SYM(CommCount, "ncclCommCount") MS_SYM(Send, "ncclSend") MS_SYM(Recv, "ncclRecv")
        MS_SYM(Broadcast, "ncclBroadcast") MS_SYM(AllGather, "ncclAllGather") MS_SYM(GroupStart, "ncclGroupStart") MS_SYM(GroupEnd, "ncclGroupEnd")
        MS_SYM(GetErrorString, "ncclGetErrorString")
#undef MS_SYM
        r.ok = true;
        return r;
    }();
    return R;
}
#define MS_NCCL(expr)                                                                                                        \
    do { const int r_ = (expr); if (r_ != ncclSuccess) return ::ms::fail(MS_ERR_COMM, "%s: %s (%s:%d)", #expr, rccl().GetErrorString(r_), __FILE__, __LINE__); } while (0)

// ---------------------------------------------------------------------------------------------- the id both transports start from
struct DistId {
    ncclUniqueId nccl;          // RCCL: ncclGetUniqueId; HOST: 128 random bytes (the mailbox's name is derived from them)
    int transport, nranks;
    unsigned magic, reserved;
};
static_assert(sizeof(DistId) == MS_DIST_ID_BYTES, "MS_DIST_ID_BYTES out of date");
constexpr unsigned ID_MAGIC = 0x4d534431u;       // "MSD1"

// ---------------------------------------------------------------------------------------------- HOST transport: shared-memory mailbox
constexpr size_t CHUNK = 1u << 20;               // one slot per ordered rank pair; longer messages go through it piece by piece
constexpr double TIMEOUT_S = 120.0;
struct alignas(64) Channel {
    std::atomic<unsigned long long> head;        // pieces written by the sender
    std::atomic<unsigned long long> tail;        // pieces consumed by the receiver
    unsigned long long len;                      // bytes of the piece in the slot
    unsigned char pad[64 - 3 * 8];
    unsigned char data[CHUNK];
};
struct RankInfo { int device; char pci[16]; char pad[12]; };
struct alignas(64) ShmHeader {
    std::atomic<unsigned> magic;
    unsigned nranks;
    std::atomic<unsigned> attached;
    std::atomic<unsigned> bar_count;
    std::atomic<unsigned> bar_gen;
    RankInfo info[MS_DIST_MAX_RANKS];
};
size_t shm_bytes(int n) { return sizeof(ShmHeader) + sizeof(Channel) * (size_t)n * n; }

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Backoff {            // spin, then yield, then sleep; gives up after TIMEOUT_S
    int n = 0;
    double t0 = now_s();
    bool wait()
    {
        if (++n < 200) return true;
        if (n < 2000) { sched_yield(); return true; }
        usleep(100);
        return (n & 1023) != 0 || now_s() - t0 < TIMEOUT_S;
    }
};

struct Op {                 // one point-to-point transfer of a group
    bool send;
    unsigned char *buf;
    size_t bytes, done = 0;
    int peer, mem;
    hipStream_t st;
    bool finished = false;  // every byte moved (an empty message: its one empty piece moved)
};

}  // namespace
}  // namespace ms

struct ms_dist {
    int rank = 0, nranks = 1, transport = MS_DIST_HOST, device = 0;
    ms_dist_info info{};
    // RCCL
    ms::ncclComm_t comm = nullptr;
    void *stage = nullptr;            // device staging for host-memory payloads (headers, meshes)
    size_t stage_cap = 0;
    std::vector<void *> stage_retired;   // outgrown staging buffers, released by ms_dist_destroy: hipFree synchronises the whole device, and a rank must not wait for
                                         // the device inside a collective (its peers' transfers may be waiting for this rank's half)
    // HOST
    ms::ShmHeader *shm = nullptr;
    size_t shm_len = 0;
    char shm_name[48] = {0};          // rank 0, while the name still exists in /dev/shm (unlinked as soon as every rank has mapped it, or by ms_dist_destroy on a failed attach)
    ms::Channel *chan(int src, int dst) const { return reinterpret_cast<ms::Channel *>(reinterpret_cast<unsigned char *>(shm) + sizeof(ms::ShmHeader)) + ((size_t)src * nranks + dst); }
    // group state
    bool grouping = false;
    std::vector<ms::Op> ops;
};

namespace ms {
namespace {

// Device buffers move on the operation's OWN stream and the host waits for that stream: the data is then ordered with what the caller enqueues next on it
// (a plain hipMemcpy runs on the NULL stream, which a non-blocking stream does not wait for) and the mailbox slot is free / filled when the call returns.
int copy_in(void *dst_host, const void *src, size_t n, int mem, hipStream_t st)      // user buffer -> mailbox slot
{
    if (n == 0) return MS_OK;
    if (mem == MS_DIST_MEM_HOST) { memcpy(dst_host, src, n); return MS_OK; }
    MS_HIP(hipMemcpyAsync(dst_host, src, n, hipMemcpyDeviceToHost, st));
    MS_HIP(hipStreamSynchronize(st));
    return MS_OK;
}
int copy_out(void *dst, const void *src_host, size_t n, int mem, hipStream_t st)     // mailbox slot -> user buffer
{
    if (n == 0) return MS_OK;
    if (mem == MS_DIST_MEM_HOST) { memcpy(dst, src_host, n); return MS_OK; }
    MS_HIP(hipMemcpyAsync(dst, src_host, n, hipMemcpyHostToDevice, st));
    MS_HIP(hipStreamSynchronize(st));
    return MS_OK;
}

// advance one operation by at most one piece; *moved says whether anything happened
int host_step(ms_dist *d, Op &op, bool *moved)
{
    *moved = false;
    if (op.finished) return MS_OK;
    Channel *c = op.send ? d->chan(d->rank, op.peer) : d->chan(op.peer, d->rank);
    const unsigned long long h = c->head.load(std::memory_order_acquire), t = c->tail.load(std::memory_order_acquire);
    if (op.send) {
        if (h != t) return MS_OK;                                   // the slot still holds a piece the peer has not taken
        const size_t n = std::min(CHUNK, op.bytes - op.done);
        if (int e = copy_in(c->data, op.buf + op.done, n, op.mem, op.st)) return e;
        c->len = n;
        c->head.store(h + 1, std::memory_order_release);
        op.done += n;
    } else {
        if (h == t) return MS_OK;                                   // nothing there yet
        const size_t n = (size_t)c->len;
        if (n > op.bytes - op.done) return fail(MS_ERR_COMM, "ms_dist: rank %d receives %zu bytes from rank %d but %zu were posted (mismatched send / recv sizes)", d->rank, n, op.peer, op.bytes - op.done);
        if (int e = copy_out(op.buf + op.done, c->data, n, op.mem, op.st)) return e;
        c->tail.store(t + 1, std::memory_order_release);
        op.done += n;
    }
    *moved = true;
    op.finished = op.done == op.bytes;          // (an empty message is one empty piece)
    return MS_OK;
}

int host_run(ms_dist *d, std::vector<Op> &ops)
{
    // every stream the buffers were produced on has to be idle before the host touches them
    for (Op &op : ops) if (op.mem == MS_DIST_MEM_DEVICE) MS_HIP(hipStreamSynchronize(op.st));
    Backoff bo;
    for (;;) {
        bool all = true, any = false;
        for (size_t i = 0; i < ops.size(); ++i) {
            Op &op = ops[i];
            if (op.finished) continue;
            // One untagged single-slot channel per ordered rank pair: the operations of a group that share a channel -- same peer, same direction -- run strictly in
            // posting order, the next one only after the previous one's last piece (ncclGroup semantics for matching sends and receives posted in the same order on
            // both sides).  Advancing them round-robin would let a second receive take the first message's pieces, or interleave the pieces of two sends.
            bool blocked = false;
            for (size_t j = 0; j < i && !blocked; ++j) blocked = !ops[j].finished && ops[j].peer == op.peer && ops[j].send == op.send;
            if (blocked) { all = false; continue; }
            bool moved = false;
            if (int e = host_step(d, op, &moved)) return e;
            any |= moved;
            all &= op.finished;
        }
        if (all) return MS_OK;
        if (any) { bo = Backoff(); continue; }
        if (!bo.wait()) return fail(MS_ERR_COMM, "ms_dist: rank %d timed out after %.0f s waiting for a peer (host transport)", d->rank, TIMEOUT_S);
    }
}

int host_barrier(ms_dist *d)
{
    ShmHeader *H = d->shm;
    const unsigned gen = H->bar_gen.load(std::memory_order_acquire);
    if (H->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)d->nranks) {
        H->bar_count.store(0, std::memory_order_relaxed);
        H->bar_gen.store(gen + 1, std::memory_order_release);
        return MS_OK;
    }
    Backoff bo;
    while (H->bar_gen.load(std::memory_order_acquire) == gen)
        if (!bo.wait()) return fail(MS_ERR_COMM, "ms_dist: rank %d timed out in a barrier (host transport)", d->rank);
    return MS_OK;
}

void shm_name(const DistId &id, char out[48])
{
    unsigned long long a = 0, b = 0;
    memcpy(&a, id.nccl.internal, 8); memcpy(&b, id.nccl.internal + 8, 8);
    snprintf(out, 48, "/msdist_%016llx%016llx", a, b);
}

int host_attach(ms_dist *d, const DistId &id)
{
    char name[48];
    shm_name(id, name);
    const size_t len = shm_bytes(d->nranks);
    int fd = -1;
    if (d->rank == 0) {
        shm_unlink(name);                                            // (a stale segment of a crashed run with the same id: impossible in practice, harmless)
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0) memcpy(d->shm_name, name, sizeof(name));        // from here on every failure path ends in ms_dist_destroy, which unlinks it
        if (fd < 0 || ftruncate(fd, (off_t)len) != 0) { if (fd >= 0) close(fd); return fail(MS_ERR_COMM, "ms_dist: cannot create the shared-memory mailbox %s (%zu bytes)", name, len); }
    } else {
        Backoff bo;
        for (;;) {
            fd = shm_open(name, O_RDWR, 0600);
            struct stat sb;
            if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size == len) break;
            if (fd >= 0) { close(fd); fd = -1; }
            if (!bo.wait()) return fail(MS_ERR_COMM, "ms_dist: rank %d never saw rank 0's mailbox %s", d->rank, name);
        }
    }
    void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail(MS_ERR_COMM, "ms_dist: mmap of the mailbox failed");
    d->shm = static_cast<ShmHeader *>(p);
    d->shm_len = len;
    ShmHeader *H = d->shm;
    if (d->rank == 0) {                                              // a fresh tmpfs segment is zero-filled: heads, tails, counters start at 0
        H->nranks = (unsigned)d->nranks;
        H->magic.store(ID_MAGIC, std::memory_order_release);
    } else {
        Backoff bo;
        while (H->magic.load(std::memory_order_acquire) != ID_MAGIC)
            if (!bo.wait()) return fail(MS_ERR_COMM, "ms_dist: rank %d: mailbox never initialised", d->rank);
        if (H->nranks != (unsigned)d->nranks) return fail(MS_ERR_COMM, "ms_dist: rank %d joined a mailbox of %u ranks, expected %d", d->rank, H->nranks, d->nranks);
    }
    RankInfo &me = H->info[d->rank];
    me.device = d->device;
    memset(me.pci, 0, sizeof(me.pci));
    if (d->device >= 0) (void)hipDeviceGetPCIBusId(me.pci, (int)sizeof(me.pci), d->device);
    H->attached.fetch_add(1, std::memory_order_acq_rel);
    Backoff bo;
    while (H->attached.load(std::memory_order_acquire) < (unsigned)d->nranks)
        if (!bo.wait()) return fail(MS_ERR_COMM, "ms_dist: rank %d: only %u of %d ranks joined within %.0f s", d->rank, H->attached.load(), d->nranks, TIMEOUT_S);
    if (int e = host_barrier(d)) return e;                           // everybody has mapped the segment ...
    if (d->rank == 0) { shm_unlink(name); d->shm_name[0] = 0; }      // ... so the name can go: nothing is left behind whatever happens later
    for (int r = 0; r < d->nranks; ++r) { d->info.device[r] = H->info[r].device; memcpy(d->info.pci_bus_id[r], H->info[r].pci, 16); }
    return MS_OK;
}

int stage_get(ms_dist *d, size_t n, void **out)
{
    if (n > d->stage_cap) {
        if (d->stage) d->stage_retired.push_back(d->stage);
        d->stage = nullptr; d->stage_cap = 0;
        const size_t want = std::max<size_t>((n + 65535) & ~(size_t)65535, (size_t)1 << 20);
        MS_HIP(hipMalloc(&d->stage, want));
        d->stage_cap = want;
    }
    *out = d->stage;
    return MS_OK;
}

int rccl_attach(ms_dist *d, const DistId &id)
{
    Rccl &R = rccl();
    if (!R.ok) return fail(MS_ERR_COMM, "ms_dist: RCCL transport requested but %s", R.why);
    MS_HIP(hipSetDevice(d->device));
    MS_NCCL(R.CommInitRank(&d->comm, d->nranks, id.nccl, d->rank));
    MS_NCCL(R.GetVersion(&d->info.rccl_version));
    // the prototypes above are hand-declared from rccl.h 2.27.7 (ROCm 7.2; csrc/rccl_abi_check.cpp holds them against the installed header at build time): NCCL keeps
    // these entry points stable within a major version, so refuse anything else rather than call through a changed ABI
    if (d->info.rccl_version / 10000 != 2) return fail(MS_ERR_COMM, "ms_dist: librccl reports version %d; the declarations in dist.cpp are for major version 2", d->info.rccl_version);
    MS_NCCL(R.CommCount(d->comm, &d->info.comm_nranks));
    // device ordinal + PCI bus id of every rank: one small all-gather (also the first real traffic on the communicator)
    RankInfo me{};
    me.device = d->device;
    (void)hipDeviceGetPCIBusId(me.pci, (int)sizeof(me.pci), d->device);
    void *st;
    if (int e = stage_get(d, sizeof(RankInfo) * (size_t)(d->nranks + 1), &st)) return e;
    RankInfo *dev_all = static_cast<RankInfo *>(st), *dev_me = dev_all + d->nranks;
    // on a stream of its own: the NULL stream is shared by every thread of a process that drives this device, and it orders against every blocking stream
    hipStream_t s0;
    MS_HIP(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    std::vector<RankInfo> all((size_t)d->nranks);
    int err = MS_OK;
    do {
        if (hipMemcpyAsync(dev_me, &me, sizeof(me), hipMemcpyHostToDevice, s0) != hipSuccess || hipStreamSynchronize(s0) != hipSuccess) { err = fail(MS_ERR_HIP, "ms_dist: staging the rank record failed"); break; }
        const int r_ = R.AllGather(dev_me, dev_all, sizeof(RankInfo), ncclUint8, d->comm, s0);
        if (r_ != ncclSuccess) { err = fail(MS_ERR_COMM, "ncclAllGather at creation: %s", R.GetErrorString(r_)); break; }
        if (hipMemcpyAsync(all.data(), dev_all, sizeof(RankInfo) * (size_t)d->nranks, hipMemcpyDeviceToHost, s0) != hipSuccess || hipStreamSynchronize(s0) != hipSuccess) { err = fail(MS_ERR_HIP, "ms_dist: reading the rank records failed"); break; }
    } while (0);
    (void)hipStreamDestroy(s0);
    if (err) return err;
    for (int r = 0; r < d->nranks; ++r) { d->info.device[r] = all[(size_t)r].device; memcpy(d->info.pci_bus_id[r], all[(size_t)r].pci, 16); }
    return MS_OK;
}

int check_peer(const ms_dist *d, int peer, const char *what)
{
    if (!d) return fail(MS_ERR_INVALID, "%s: null ms_dist", what);
    if (peer < 0 || peer >= d->nranks) return fail(MS_ERR_INVALID, "%s: peer %d out of range (nranks %d)", what, peer, d->nranks);
    return MS_OK;
}

int p2p(ms_dist *d, bool send, void *buf, size_t bytes, int peer, int mem, hipStream_t st, const char *what)
{
    if (int e = check_peer(d, peer, what)) return e;
    MS_CHECK(buf || bytes == 0, "%s: null buffer", what);
    MS_CHECK(mem == MS_DIST_MEM_DEVICE || mem == MS_DIST_MEM_HOST, "%s: bad memory kind %d", what, mem);
    MS_CHECK(peer != d->rank || d->grouping, "%s: a transfer to oneself needs its counterpart in the same group", what);
    if (mem == MS_DIST_MEM_DEVICE && d->device < 0) return fail(MS_ERR_NO_DEVICE, "%s: this communicator was created without a device: host-memory messages only", what);
    if (d->transport == MS_DIST_RCCL) {
        Rccl &R = rccl();
        if (mem == MS_DIST_MEM_HOST) return fail(MS_ERR_UNSUPPORTED, "%s: host memory goes through ms_dist_broadcast / ms_dist_mesh_exchange on the RCCL transport", what);
        if (send) MS_NCCL(R.Send(buf, bytes, ncclUint8, peer, d->comm, st));
        else MS_NCCL(R.Recv(buf, bytes, ncclUint8, peer, d->comm, st));
        return MS_OK;
    }
    Op op{send, static_cast<unsigned char *>(buf), bytes, 0, peer, mem, st};
    if (d->grouping) { d->ops.push_back(op); return MS_OK; }
    std::vector<Op> one{op};
    return host_run(d, one);
}

}  // namespace
}  // namespace ms

using namespace ms;

extern "C" {

// The file the RCCL entry points were resolved from (dladdr of ncclCommInitRank): what a first multi-GPU run prints BEFORE its timed regions, so that "which librccl
// did the product load next to PyTorch's own copy" is a fact of the record, not a guess.  Resolves RCCL if that has not happened yet; "" (MS_ERR_COMM) when none loads.
int ms_dist_rccl_library_path(char *out, size_t cap)
{
    if (!out || cap == 0) return fail(MS_ERR_INVALID, "ms_dist_rccl_library_path: null buffer");
    out[0] = 0;
    Rccl &R = rccl();
    if (!R.ok) return fail(MS_ERR_COMM, "RCCL is not available: %s", R.why);
    Dl_info di{};
    if (!dladdr(reinterpret_cast<void *>(R.CommInitRank), &di) || !di.dli_fname) return fail(MS_ERR_COMM, "dladdr found no file for ncclCommInitRank");
    char real[4096];
    const char *name = realpath(di.dli_fname, real) ? real : di.dli_fname;
    snprintf(out, cap, "%s", name);
    return MS_OK;
}

int ms_dist_set_rccl_library(const char *path)
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl_resolved) return fail(MS_ERR_STATE, "ms_dist_set_rccl_library: RCCL has already been resolved in this process (call it before the first RCCL communicator or id)");
    g_rccl_path = path ? path : "";
    return MS_OK;
}

int ms_dist_unique_id(int transport, int nranks, void *id_out)
{
    MS_CHECK(id_out && nranks >= 1 && nranks <= MS_DIST_MAX_RANKS, "ms_dist_unique_id: nranks %d not in [1, %d]", nranks, MS_DIST_MAX_RANKS);
    MS_CHECK(transport == MS_DIST_AUTO || transport == MS_DIST_RCCL || transport == MS_DIST_HOST, "ms_dist_unique_id: bad transport %d", transport);
    DistId id{};
    if (transport == MS_DIST_AUTO) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
        transport = (ndev >= nranks && ndev >= 1 && rccl().ok) ? MS_DIST_RCCL : MS_DIST_HOST;      // fewer devices than ranks: ranks share a GPU, which RCCL refuses
    }
    if (transport == MS_DIST_RCCL) {
        Rccl &R = rccl();
        if (!R.ok) return fail(MS_ERR_COMM, "ms_dist_unique_id: RCCL transport requested but %s", R.why);
        MS_NCCL(R.GetUniqueId(&id.nccl));
    } else {
        FILE *f = fopen("/dev/urandom", "rb");
        const size_t got = f ? fread(id.nccl.internal, 1, sizeof(id.nccl.internal), f) : 0;
        if (f) fclose(f);
        if (got != sizeof(id.nccl.internal)) {
            unsigned long long x = (unsigned long long)getpid() * 0x9e3779b97f4a7c15ull ^ (unsigned long long)(now_s() * 1e9);
            for (size_t i = 0; i < sizeof(id.nccl.internal); ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; id.nccl.internal[i] = (char)x; }
        }
    }
    id.transport = transport; id.nranks = nranks; id.magic = ID_MAGIC;
    memcpy(id_out, &id, sizeof(id));
    return MS_OK;
}

int ms_dist_create(ms_dist **out, int rank, int nranks, const void *id_bytes, int device)
{
    MS_CHECK(out && id_bytes, "ms_dist_create: null argument");
    *out = nullptr;
    DistId id;
    memcpy(&id, id_bytes, sizeof(id));
    MS_CHECK(id.magic == ID_MAGIC, "ms_dist_create: not an id from ms_dist_unique_id");
    MS_CHECK(nranks == id.nranks && rank >= 0 && rank < nranks, "ms_dist_create: rank %d / nranks %d do not match the id (made for %d ranks)", rank, nranks, id.nranks);
    ms_dist *d = new ms_dist();
    d->rank = rank; d->nranks = nranks; d->transport = id.transport; d->device = device;
    d->info.rank = rank; d->info.nranks = nranks; d->info.transport = id.transport;
    for (int r = 0; r < MS_DIST_MAX_RANKS; ++r) d->info.device[r] = -1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    if (device >= 0 && device < ndev) (void)hipSetDevice(device);
    else if (id.transport == MS_DIST_RCCL || (ndev > 0 && device >= 0)) { delete d; return fail(MS_ERR_NO_DEVICE, "ms_dist_create: device %d not present (%d visible)", device, ndev); }
    else d->device = -1;                                             // host transport with no GPU (or device < 0): host-memory messages only, MS_DIST_MEM_DEVICE is refused
    const int e = id.transport == MS_DIST_RCCL ? rccl_attach(d, id) : host_attach(d, id);
    if (e) { ms_dist_destroy(d); return e; }
    *out = d;
    return MS_OK;
}

void ms_dist_destroy(ms_dist *d)
{
    if (!d) return;
    if (d->comm) (void)rccl().CommDestroy(d->comm);
    if (d->stage) (void)hipFree(d->stage);
    for (void *p : d->stage_retired) (void)hipFree(p);
    if (d->shm) munmap(d->shm, d->shm_len);
    if (d->shm_name[0]) shm_unlink(d->shm_name);                     // rank 0 of an attach that failed (a peer never joined, a barrier timed out): nranks^2 MiB would stay in /dev/shm
    delete d;
}

int ms_dist_get_info(const ms_dist *d, ms_dist_info *info)
{
    MS_CHECK(d && info, "ms_dist_get_info: null argument");
    *info = d->info;
    return MS_OK;
}

int ms_dist_send(ms_dist *d, const void *buf, size_t bytes, int peer, int mem, ms_stream stream)
{
    return p2p(d, true, const_cast<void *>(buf), bytes, peer, mem, as_stream(stream), "ms_dist_send");
}
int ms_dist_recv(ms_dist *d, void *buf, size_t bytes, int peer, int mem, ms_stream stream)
{
    return p2p(d, false, buf, bytes, peer, mem, as_stream(stream), "ms_dist_recv");
}

int ms_dist_group_begin(ms_dist *d)
{
    MS_CHECK(d && !d->grouping, "ms_dist_group_begin: null or already in a group");
    d->grouping = true;
    d->ops.clear();
    if (d->transport == MS_DIST_RCCL) MS_NCCL(rccl().GroupStart());
    return MS_OK;
}
int ms_dist_group_end(ms_dist *d)
{
    MS_CHECK(d && d->grouping, "ms_dist_group_end: no group open");
    d->grouping = false;
    if (d->transport == MS_DIST_RCCL) { MS_NCCL(rccl().GroupEnd()); return MS_OK; }
    // a send to oneself pairs with the receive from oneself of the same group: a plain copy
    std::vector<Op> ops;
    ops.swap(d->ops);
    for (size_t i = 0; i < ops.size(); ++i) {
        if (ops[i].peer != d->rank || !ops[i].send || ops[i].finished) continue;
        for (size_t j = 0; j < ops.size(); ++j)
            if (!ops[j].send && ops[j].peer == d->rank && !ops[j].finished && ops[j].bytes == ops[i].bytes) {
                if (ops[i].mem == MS_DIST_MEM_HOST && ops[j].mem == MS_DIST_MEM_HOST) memcpy(ops[j].buf, ops[i].buf, ops[i].bytes);
                else {      // behind what produced the source, on the receive's stream, complete on return (as the mailbox path)
                    if (ops[i].mem == MS_DIST_MEM_DEVICE && ops[i].st != ops[j].st) MS_HIP(hipStreamSynchronize(ops[i].st));
                    MS_HIP(hipMemcpyAsync(ops[j].buf, ops[i].buf, ops[i].bytes, hipMemcpyDefault, ops[j].st));
                    MS_HIP(hipStreamSynchronize(ops[j].st));
                }
                ops[i].finished = ops[j].finished = true;
                break;
            }
        if (!ops[i].finished) return fail(MS_ERR_INVALID, "ms_dist_group_end: a send to oneself has no matching receive in the group");
    }
    std::vector<Op> rest;
    for (const Op &o : ops) {
        if (o.finished) continue;
        if (o.peer == d->rank) return fail(MS_ERR_INVALID, "ms_dist_group_end: a receive from oneself has no matching send in the group");
        rest.push_back(o);
    }
    return host_run(d, rest);
}

int ms_dist_broadcast(ms_dist *d, void *buf, size_t bytes, int root, int mem, ms_stream stream)
{
    if (int e = check_peer(d, root, "ms_dist_broadcast")) return e;
    MS_CHECK(buf && bytes > 0 && !d->grouping, "ms_dist_broadcast: null buffer, empty message or inside a group");
    hipStream_t st = as_stream(stream);
    if (d->transport == MS_DIST_RCCL) {
        Rccl &R = rccl();
        if (mem == MS_DIST_MEM_DEVICE) { MS_NCCL(R.Broadcast(buf, buf, bytes, ncclUint8, root, d->comm, st)); return MS_OK; }
        void *dev;                                                   // host payload (headers, vertex meshes: tens of KB): staged, synchronous
        if (int e = stage_get(d, bytes, &dev)) return e;
        if (d->rank == root) MS_HIP(hipMemcpyAsync(dev, buf, bytes, hipMemcpyHostToDevice, st));
        MS_NCCL(R.Broadcast(dev, dev, bytes, ncclUint8, root, d->comm, st));
        if (d->rank != root) MS_HIP(hipMemcpyAsync(buf, dev, bytes, hipMemcpyDeviceToHost, st));
        MS_HIP(hipStreamSynchronize(st));
        return MS_OK;
    }
    std::vector<Op> ops;
    if (d->rank == root) { for (int r = 0; r < d->nranks; ++r) if (r != root) ops.push_back(Op{true, static_cast<unsigned char *>(buf), bytes, 0, r, mem, st}); }
    else ops.push_back(Op{false, static_cast<unsigned char *>(buf), bytes, 0, root, mem, st});
    return ops.empty() ? MS_OK : host_run(d, ops);
}

int ms_dist_barrier(ms_dist *d, ms_stream stream)
{
    MS_CHECK(d && !d->grouping, "ms_dist_barrier: null or inside a group");
    hipStream_t st = as_stream(stream);
    if (d->transport == MS_DIST_RCCL) {
        void *dev;
        if (int e = stage_get(d, 4 * (size_t)(d->nranks + 1), &dev)) return e;
        unsigned char *b = static_cast<unsigned char *>(dev);
        MS_NCCL(rccl().AllGather(b + 4 * (size_t)d->nranks, b, 4, ncclUint8, d->comm, st));
        MS_HIP(hipStreamSynchronize(st));
        return MS_OK;
    }
    if (d->device >= 0) MS_HIP(hipStreamSynchronize(st));
    return host_barrier(d);
}

int ms_dist_gather_slabs(ms_dist *d, const void *slab, size_t bytes, void *const *recv, int sink, ms_stream stream)
{
    if (int e = check_peer(d, sink, "ms_dist_gather_slabs")) return e;
    MS_CHECK(bytes > 0 && !d->grouping, "ms_dist_gather_slabs: empty slab or inside a group");
    if (d->nranks == 1) return MS_OK;
    // A rank that finds its own arguments wrong still takes part in the collective (into / out of scratch) and reports the error afterwards: leaving early would
    // leave the peers blocked in their half of the transfer (120 s on the host transport, for ever on RCCL) and the communicator out of step.
    int bad = MS_OK;
    void *scratch = nullptr;
    if (d->rank == sink) {
        bool missing = !recv;
        for (int r = 0; r < d->nranks && !missing; ++r) missing = r != sink && !recv[r];
        if (missing) bad = fail(MS_ERR_INVALID, "ms_dist_gather_slabs: the sink needs one receive buffer per peer");
    } else if (!slab) bad = fail(MS_ERR_INVALID, "ms_dist_gather_slabs: null slab");
    if (bad) {
        if (d->device < 0) return bad;                    // (no device at all: nothing can be posted; device slabs are refused on such a communicator anyway)
        if (int e = stage_get(d, bytes, &scratch)) return e;
    }
    if (int e = ms_dist_group_begin(d)) return e;
    int err = MS_OK;
    if (d->rank == sink) {
        for (int r = 0; r < d->nranks && !err; ++r) {
            if (r == sink) continue;
            err = ms_dist_recv(d, (recv && recv[r]) ? recv[r] : scratch, bytes, r, MS_DIST_MEM_DEVICE, stream);
        }
    } else err = ms_dist_send(d, slab ? slab : scratch, bytes, sink, MS_DIST_MEM_DEVICE, stream);
    if (err) {                  // close the group without running it (RCCL: end the group so the communicator stays usable), report the first error
        d->grouping = false;
        d->ops.clear();
        if (d->transport == MS_DIST_RCCL) (void)rccl().GroupEnd();
        return err;
    }
    if (int e = ms_dist_group_end(d)) return e;
    if (bad) { if (d->device >= 0) (void)hipStreamSynchronize(as_stream(stream)); return fail(MS_ERR_INVALID, "ms_dist_gather_slabs: rank %d had a null buffer (the transfer ran into scratch memory so that the peers return)", d->rank); }
    return MS_OK;
}

namespace {
struct MeshHeader { unsigned magic; int have, version, n_views, rows, cols; long long swap_frame; };      // have: 0 no update, 1 update follows, -1 the root's update was invalid
}

int ms_dist_mesh_exchange(ms_dist *d, int root, const ms_dist_mesh_update *upd, ms_dist_mesh_update *out, size_t cap_floats, int *have, ms_stream stream)
{
    if (int e = check_peer(d, root, "ms_dist_mesh_exchange")) return e;
    if (have) *have = 0;
    // Every rank takes part in the header broadcast whatever it thinks of its own arguments, and in the payload broadcast whenever the header announces one: the
    // status travels in the header (root's update invalid) or is decided from it identically everywhere, so no rank leaves the collective half way.
    const bool out_ok = out && have && out->mesh_x && out->mesh_y;
    MeshHeader h{ID_MAGIC, 0, 0, 0, 0, 0, 0};
    if (d->rank == root && upd) {
        const bool ok = upd->mesh_x && upd->mesh_y && upd->n_views >= 1 && upd->n_views <= 16 && upd->rows >= 2 && upd->cols >= 2;
        h.have = ok ? 1 : -1;
        if (ok) { h.version = upd->version; h.n_views = upd->n_views; h.rows = upd->rows; h.cols = upd->cols; h.swap_frame = upd->swap_frame; }
    }
    if (d->nranks > 1) if (int e = ms_dist_broadcast(d, &h, sizeof(h), root, MS_DIST_MEM_HOST, stream)) return e;
    if (h.magic != ID_MAGIC) return fail(MS_ERR_COMM, "ms_dist_mesh_exchange: rank %d received a corrupt header (ranks out of step?)", d->rank);
    if (h.have < 0) return fail(MS_ERR_INVALID, "ms_dist_mesh_exchange: bad update on rank %d (every rank returns this)", root);
    if (!h.have) { MS_CHECK(out_ok, "ms_dist_mesh_exchange: null output"); return MS_OK; }
    const size_t n = (size_t)h.n_views * h.rows * h.cols;
    std::vector<float> pack(2 * n);
    if (d->rank == root) { memcpy(pack.data(), upd->mesh_x, n * sizeof(float)); memcpy(pack.data() + n, upd->mesh_y, n * sizeof(float)); }
    if (d->nranks > 1) if (int e = ms_dist_broadcast(d, pack.data(), 2 * n * sizeof(float), root, MS_DIST_MEM_HOST, stream)) return e;
    MS_CHECK(out_ok, "ms_dist_mesh_exchange: null output");
    MS_CHECK(n <= cap_floats, "ms_dist_mesh_exchange: update of %zu floats per map exceeds the caller's capacity %zu", n, cap_floats);
    memcpy(out->mesh_x, pack.data(), n * sizeof(float));
    memcpy(out->mesh_y, pack.data() + n, n * sizeof(float));
    out->swap_frame = h.swap_frame; out->version = h.version; out->n_views = h.n_views; out->rows = h.rows; out->cols = h.cols;
    *have = 1;
    return MS_OK;
}

int ms_dist_apply_meshes(ms_ctx *ctx, const ms_dist_mesh_update *upd, long long next_frame, int *applied, ms_stream stream)
{
    MS_CHECK(ctx && upd && applied, "ms_dist_apply_meshes: null argument");
    *applied = 0;
    if (next_frame < upd->swap_frame) return MS_OK;
    // convertMeshesToMap for every view in one call (two launches): the update carries the meshes of all views back to back
    ms_view_geom g;
    MS_CHECK(upd->n_views >= 1 && ms_get_view_geom(ctx, upd->n_views - 1, &g) == MS_OK && ms_get_view_geom(ctx, upd->n_views, &g) != MS_OK,
             "ms_dist_apply_meshes: the update holds %d meshes, not one per view of the context", upd->n_views);
    if (int e = ms_set_meshes(ctx, upd->mesh_x, upd->mesh_y, upd->rows, upd->cols, stream)) return e;
    *applied = 1;
    return MS_OK;
}

}  // extern "C"
