/*
 * ms_dist.h -- multi-GPU layer of libmsstitch.so: one rank (thread or process) per MI355X, RCCL over xGMI for the only exchanges the
 * path has:
 *   - frame-parallel stitching: frame t -> rank t mod G, every rank a replica of the static tables, finished equirect slabs (planar I420,
 *     the encoder input of consume(), APP/timed.cpp:308-316) sent to the sink rank with ncclSend / ncclRecv;
 *   - recalibration (APP/timed.cpp:414-463 recalibrateMesh, meshwarper.cpp:879-884): the rank that solved the N x M CPW meshes broadcasts
 *     them (ncclBroadcast) together with the frame index at which EVERY rank swaps them in, so a G-GPU run is bit-identical to 1 GPU;
 *   - pano-column / view shards of one frame (BASELINE configs[4]): column slabs or int16 partial sums to the group's first rank.
 * The reference is single-device (timed.cpp:496 cuda::setDevice(0)): nothing here replaces a reference interface; it is what a maintainer
 * calls around ms_stitch when the thread / queue graph of timed.cpp is instantiated once per GPU.
 *
 * Transports.  MS_DIST_RCCL: librccl.so.1 (dlopen'ed on first use: single-GPU users never load it), one communicator, device buffers,
 * stream-ordered, ncclGroup for concurrent send/recv.  MS_DIST_HOST: ranks that SHARE a device (RCCL refuses two ranks on one GPU) or have
 * no peer access exchange through a POSIX shared-memory mailbox -- blocking, staged through the host, for tests and single-GPU boxes; it
 * accepts host pointers too (MS_DIST_MEM_HOST), so the protocol is testable without a GPU.  Same calls, same results.
 *
 * Threading: one ms_dist per rank, used by one thread at a time.  Errors: ms_status codes of ms_stitch.h + ms_last_error().
 */
#ifndef MS_DIST_H
#define MS_DIST_H

#include "ms_stitch.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { MS_DIST_AUTO = 0, MS_DIST_RCCL = 1, MS_DIST_HOST = 2 };
enum { MS_DIST_MEM_DEVICE = 0, MS_DIST_MEM_HOST = 1 };
#define MS_DIST_ID_BYTES 144          /* ncclUniqueId (128) + transport, nranks, magic, reserved */
#define MS_DIST_MAX_RANKS 16

typedef struct ms_dist ms_dist;

typedef struct ms_dist_info {
    int rank, nranks;
    int transport;                       /* MS_DIST_RCCL / MS_DIST_HOST (AUTO resolved) */
    int rccl_version;                    /* ncclGetVersion(), 0 for the host transport */
    int comm_nranks;                     /* ncclCommCount() of the communicator: what RCCL itself saw (= nranks), 0 for the host transport */
    int device[MS_DIST_MAX_RANKS];       /* HIP device ordinal of every rank (all-gathered at creation) */
    char pci_bus_id[MS_DIST_MAX_RANKS][16];   /* hipDeviceGetPCIBusId of every rank: distinct ids = distinct GPUs */
} ms_dist_info;

/* Where the RCCL transport finds RCCL.  By default librccl.so.1 is resolved on first use: a copy already loaded into the process (PyTorch ships its own under the
 * same soname) is reused, otherwise the loader's search path, then /opt/rocm/lib.  A deployment that keeps RCCL elsewhere -- or a test that substitutes a loopback
 * implementation of the same entry points (tests/fake_rccl.cpp) -- names the file here, once per process, BEFORE the first RCCL id / communicator
 * (MS_ERR_STATE afterwards; NULL or "" restores the default).  The library reads no environment variable for this. */
MS_API int ms_dist_set_rccl_library(const char *path);
/* The file the RCCL entry points resolved from (realpath of dladdr(ncclCommInitRank)), NUL-terminated into out[cap]; resolves RCCL on first use.  bench.py --gpus N and
 * stitch_dist print it before their first timed region: with PyTorch in the process two copies of librccl exist on a ROCm box, and the record should say which one ran. */
MS_API int ms_dist_rccl_library_path(char *out, size_t cap);

/* Rank 0 calls this once and hands the bytes to every rank (file, pipe, torch.distributed store, a shared variable between threads).
 * transport AUTO: RCCL when this process sees at least `nranks` devices, else HOST. */
MS_API int ms_dist_unique_id(int transport, int nranks, void *id_out /* MS_DIST_ID_BYTES */);
/* Collective over all ranks (blocks until every rank has joined; 120 s timeout).  `device` = the HIP device this rank drives (made current). */
MS_API int ms_dist_create(ms_dist **out, int rank, int nranks, const void *id, int device);
MS_API void ms_dist_destroy(ms_dist *d);
MS_API int ms_dist_get_info(const ms_dist *d, ms_dist_info *info);

/* Point-to-point, ncclSend / ncclRecv semantics: RCCL: enqueued on `stream`; HOST: waits for `stream`, then blocks until the peer has taken /
 * delivered the bytes.  Between ms_dist_group_begin / _end any number of sends and receives progress together (ncclGroupStart / End). */
MS_API int ms_dist_send(ms_dist *d, const void *buf, size_t bytes, int peer, int mem, ms_stream stream);
MS_API int ms_dist_recv(ms_dist *d, void *buf, size_t bytes, int peer, int mem, ms_stream stream);
MS_API int ms_dist_group_begin(ms_dist *d);
MS_API int ms_dist_group_end(ms_dist *d);
/* ncclBroadcast in place: root's `buf` to everyone's `buf`. */
MS_API int ms_dist_broadcast(ms_dist *d, void *buf, size_t bytes, int root, int mem, ms_stream stream);
MS_API int ms_dist_barrier(ms_dist *d, ms_stream stream);

/* The frame-parallel gather (BASELINE configs[3]): every rank hands over its contiguous slab of `bytes` bytes (n finished frames); on `sink`,
 * recv[r] (r != sink) receives rank r's slab and recv[sink] may be NULL (the sink's own frames stay where they are).  One grouped exchange. */
MS_API int ms_dist_gather_slabs(ms_dist *d, const void *slab, size_t bytes, void *const *recv, int sink, ms_stream stream);

/* ---- recalibration: broadcast of the CPW meshes + the agreed swap frame ----------------------------------------------------------------
 * Collective, called by every rank at the same batch boundary (e.g. before each ms_stitch batch while CPW is on).  On `root`, `upd` is the
 * update the recalibration thread has produced since the last call, or NULL; every rank gets `*have` = 1 and a copy of the update in `out`
 * (caller-provided arrays of capacity `cap_floats` floats each) when there was one, 0 otherwise.  swap_frame is the GLOBAL frame index from
 * which the new meshes apply on every rank (root picks a batch boundary at or after the current one): ms_dist_apply_meshes does the
 * ms_set_mesh calls when the rank's next frame index reaches it. */
typedef struct ms_dist_mesh_update {
    long long swap_frame;          /* first global frame index stitched with these meshes */
    int version;                   /* recalibration counter (monotone) */
    int n_views, rows, cols;       /* N x M vertex meshes per view (meshwarper.cpp: N rows, M columns) */
    float *mesh_x, *mesh_y;        /* host, n_views * rows * cols floats each */
} ms_dist_mesh_update;
MS_API int ms_dist_mesh_exchange(ms_dist *d, int root, const ms_dist_mesh_update *upd, ms_dist_mesh_update *out, size_t cap_floats, int *have,
                                 ms_stream stream);
/* ms_set_meshes (every view of `upd`, which must cover all views of the context) on this rank's context if `next_frame` >= upd->swap_frame; *applied = 1 then
 * (the caller drops the update). */
MS_API int ms_dist_apply_meshes(ms_ctx *ctx, const ms_dist_mesh_update *upd, long long next_frame, int *applied, ms_stream stream);

#ifdef __cplusplus
}
#endif
#endif
