/*
 * is_gather.hip -- the final gather of a sharded batch for C / C++ callers.
 *
 * The column DP has no data-path collective: images are independent, every rank (one process per GPU)
 * computes its shard, and only the stixel OUTPUT travels -- to one rank, over RCCL (xGMI on a node).
 * bench.py and the Python tests drive that through torch.distributed (instance_stixels_amd/parallel.py);
 * this file is the same exchange behind the C ABI, for the reference's C++ callers
 * (/root/reference/apps/run_cityscapes.cu:245-449 processes its frames one by one on one GPU; a multi-GPU
 * caller shards them and collects the Sections where the reference writes its .stixels files):
 *
 *   is_gather_i32       variable-size gather of int32 payloads, sizes known on the host: grouped
 *                       ncclSend / ncclRecv (rccl.h:700, 722)
 *   is_gather_sections  the compacted payload of is_pack_sections: the section totals and (for equal
 *                       shards) the per-column counts through ncclGather (rccl.h:745, an RCCL extension),
 *                       then the used sections only (10-40 of the 200 slots of a column: ~0.1-0.3 x the
 *                       bytes) through is_gather_i32
 *
 * RCCL is loaded at run time (dlopen of librccl.so.1; a process that has imported torch gets torch's
 * copy, the same soname), so the library has no link-time dependency on it and single-GPU callers never
 * touch it.  `comm` is an ncclComm_t passed as void*.
 */
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

/* The dozen RCCL names this file needs, declared here: the library dlopens RCCL at run time and must also
 * BUILD on a single-GPU installation without the RCCL headers (rccl.h:36-52, 461: an opaque communicator, a
 * 128-byte id, result 0 = success, ncclInt32 = 2; tests/test_host_and_abi.py compares these values with the
 * header where it is installed). */
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
static const ncclResult_t ncclSuccess = 0;
static const ncclDataType_t ncclInt32 = 2;
#include "instance_stixels_core.h"

extern "C" int isk_fail(int code, const char* msg);

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

void load_rccl(Rccl& r);
Rccl& rccl() { /* (two threads may make their first is_comm_* / is_gather_* call at once) */
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { load_rccl(r); });
    return r;
}
void load_rccl(Rccl& r) {
    /* IS_RCCL_LIB: another library with the same twelve entry points (read once, here).  The tests use it to run the
     * gather with more than one rank on a one-GPU box (tests/mock_rccl: shared memory + host staging; real RCCL
     * refuses two ranks on one device); a deployment could name a differently installed RCCL. */
    const char* names[] = {getenv("IS_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (n == nullptr || n[0] == 0) continue;
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle || n == names[0]) break; /* (a named library that does not load is an error, not a reason to fall back) */
    }
    if (!r.handle) return;
#define IS_SYM(field, name) \
    *(void**)(&r.field) = dlsym(r.handle, name); \
    if (!r.field) return
    IS_SYM(GetUniqueId, "ncclGetUniqueId");
    IS_SYM(CommInitRank, "ncclCommInitRank");
    IS_SYM(CommDestroy, "ncclCommDestroy");
    IS_SYM(CommCount, "ncclCommCount");
    IS_SYM(CommUserRank, "ncclCommUserRank");
    IS_SYM(Gather, "ncclGather");
    IS_SYM(Broadcast, "ncclBroadcast");
    IS_SYM(Send, "ncclSend");
    IS_SYM(Recv, "ncclRecv");
    IS_SYM(GroupStart, "ncclGroupStart");
    IS_SYM(GroupEnd, "ncclGroupEnd");
    IS_SYM(GetErrorString, "ncclGetErrorString");
#undef IS_SYM
    r.ok = true;
}

int need_rccl() {
    if (rccl().ok) return IS_OK;
    return isk_fail(IS_EHIP, "RCCL is not available: dlopen of librccl.so.1 failed or a symbol is missing "
                             "(the multi-GPU gather needs ROCm's RCCL; nothing falls back to another path)");
}

int fail_nccl(ncclResult_t e, const char* what) {
    char msg[400];
    snprintf(msg, sizeof(msg), "%s returned %s (%d)", what, rccl().GetErrorString(e), (int)e);
    return isk_fail(IS_EHIP, msg);
}
int fail_hip(hipError_t e, const char* what) {
    char msg[400];
    snprintf(msg, sizeof(msg), "%s returned %s (%d)", what, hipGetErrorString(e), (int)e);
    return isk_fail(IS_EHIP, msg);
}
#define NCCL_TRY(expr)                                         \
    do {                                                       \
        ncclResult_t e__ = (expr);                             \
        if (e__ != ncclSuccess) return fail_nccl(e__, #expr);  \
    } while (0)
#define HIPG_TRY(expr)                                         \
    do {                                                       \
        hipError_t e__ = (expr);                               \
        if (e__ != hipSuccess) return fail_hip(e__, #expr);    \
    } while (0)

int comm_shape(void* comm, int* rank, int* nranks) {
    NCCL_TRY(rccl().CommUserRank((ncclComm_t)comm, rank));
    NCCL_TRY(rccl().CommCount((ncclComm_t)comm, nranks));
    return IS_OK;
}

/* a few device words per device for the sizes / the go-ahead flag of is_gather_sections (never freed:
 * 4 KB per device a process gathers on) */
int32_t* scratch_words(int nwords) {
    static int32_t* buf[64] = {nullptr};
    static int cap[64] = {0};
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (cap[dev] < nwords) {
        int32_t* p = nullptr;
        const int n = nwords < 1024 ? 1024 : nwords;
        if (hipMalloc((void**)&p, sizeof(int32_t) * (size_t)n) != hipSuccess) return nullptr;
        buf[dev] = p; /* (an outgrown buffer is leaked on purpose: work queued on it may still be in flight) */
        cap[dev] = n;
    }
    return buf[dev];
}

}  // namespace

extern "C" {

int is_comm_unique_id(void* id_out, size_t id_bytes) {
    if (!id_out || id_bytes < sizeof(ncclUniqueId)) return isk_fail(IS_EINVAL, "invalid argument: id buffer < 128 bytes");
    if (int rc = need_rccl()) return rc;
    ncclUniqueId id;
    NCCL_TRY(rccl().GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return IS_OK;
}

int is_comm_init_rank(void** comm, int nranks, const void* id, int rank) {
    if (!comm || !id || nranks < 1 || rank < 0 || rank >= nranks) return isk_fail(IS_EINVAL, "invalid argument: is_comm_init_rank");
    if (int rc = need_rccl()) return rc;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    NCCL_TRY(rccl().CommInitRank(&c, nranks, uid, rank));
    *comm = (void*)c;
    return IS_OK;
}

int is_comm_destroy(void* comm) {
    if (!comm) return IS_OK;
    if (int rc = need_rccl()) return rc;
    NCCL_TRY(rccl().CommDestroy((ncclComm_t)comm));
    return IS_OK;
}

int is_comm_rank(void* comm, int* rank, int* nranks) {
    if (!comm || !rank || !nranks) return isk_fail(IS_EINVAL, "invalid argument: null pointer");
    if (int rc = need_rccl()) return rc;
    return comm_shape(comm, rank, nranks);
}

int is_gather_i32(void* comm, int dst, const int64_t* h_counts, const int32_t* d_send, int32_t* d_recv,
                  void* stream_) {
    if (!comm || !h_counts) return isk_fail(IS_EINVAL, "invalid argument: null pointer");
    if (int rc = need_rccl()) return rc;
    int rank = 0, nranks = 0;
    if (int rc = comm_shape(comm, &rank, &nranks)) return rc;
    if (dst < 0 || dst >= nranks) return isk_fail(IS_EINVAL, "invalid argument: dst outside the communicator");
    hipStream_t stream = (hipStream_t)stream_;
    const ncclComm_t c = (ncclComm_t)comm;
    const int64_t mine = h_counts[rank];
    if (mine < 0) return isk_fail(IS_EINVAL, "invalid argument: negative count");
    if (mine > 0 && !d_send) return isk_fail(IS_EINVAL, "invalid argument: null send buffer");
    if (rank != dst) { /* a sender only knows (needs) its own count */
        if (mine > 0) NCCL_TRY(rccl().Send(d_send, (size_t)mine, ncclInt32, dst, c, stream));
        return IS_OK;
    }
    int64_t total = 0;
    for (int r = 0; r < nranks; r++) {
        if (h_counts[r] < 0) return isk_fail(IS_EINVAL, "invalid argument: negative count");
        total += h_counts[r];
    }
    if (total > 0 && !d_recv) return isk_fail(IS_EINVAL, "invalid argument: null receive buffer on dst");
    /* (grouped, so that the receives progress together; with one rank it degenerates to the local copy) */
    NCCL_TRY(rccl().GroupStart());
    int64_t off = 0;
    ncclResult_t err = ncclSuccess;
    for (int r = 0; r < nranks && err == ncclSuccess; r++) {
        if (r != dst && h_counts[r] > 0)
            err = rccl().Recv(d_recv + off, (size_t)h_counts[r], ncclInt32, r, c, stream);
        off += h_counts[r];
    }
    const ncclResult_t end = rccl().GroupEnd();
    if (err != ncclSuccess) return fail_nccl(err, "ncclRecv");
    if (end != ncclSuccess) return fail_nccl(end, "ncclGroupEnd");
    /* dst's own part: a device-to-device copy behind the receives */
    off = 0;
    for (int r = 0; r < dst; r++) off += h_counts[r];
    if (mine > 0 && d_recv + off != d_send)
        HIPG_TRY(hipMemcpyAsync(d_recv + off, d_send, sizeof(int32_t) * (size_t)mine, hipMemcpyDeviceToDevice, stream));
    return IS_OK;
}

int is_gather_sections(void* comm, int dst, const int32_t* h_columns, const int32_t* d_counts,
                       const int32_t* d_offsets, const is_section* d_packed, int32_t* d_all_counts,
                       is_section* d_all_packed, size_t cap_sections, int64_t* h_totals, void* stream_) {
    if (!comm || !h_columns || !d_counts || !d_offsets || !h_totals)
        return isk_fail(IS_EINVAL, "invalid argument: null pointer");
    if (int rc = need_rccl()) return rc;
    int rank = 0, nranks = 0;
    if (int rc = comm_shape(comm, &rank, &nranks)) return rc;
    if (dst < 0 || dst >= nranks) return isk_fail(IS_EINVAL, "invalid argument: dst outside the communicator");
    hipStream_t stream = (hipStream_t)stream_;
    const ncclComm_t c = (ncclComm_t)comm;
    /* EVERY local check and the scratch come before the first collective: a rank that returns after it has posted
     * one leaves the others blocked in the next.  (h_columns is the same array on every rank, so its checks fail on
     * all of them alike; a failure here that is this rank's alone -- a null buffer, no scratch -- is fatal for the
     * job, as with any collective, and is reported as IS_EINVAL / IS_EHIP, never as the "grow and repeat" IS_ENOMEM
     * that only dst's broadcast go-ahead below may produce, on all ranks at once.) */
    std::vector<int64_t> cnt(nranks);
    bool equal = true; /* (all ranks take the same branch) */
    for (int r = 0; r < nranks; r++) {
        if (h_columns[r] < 0) return isk_fail(IS_EINVAL, "invalid argument: negative column count");
        cnt[r] = h_columns[r];
        equal = equal && h_columns[r] == h_columns[0];
    }
    const int my_cols = h_columns[rank];
    if (rank == dst && !d_all_counts) return isk_fail(IS_EINVAL, "invalid argument: null d_all_counts on dst");
    if (my_cols > 0 && !d_packed) return isk_fail(IS_EINVAL, "invalid argument: null d_packed");
    int32_t* d_words = scratch_words(nranks + 1);
    if (!d_words) return isk_fail(IS_EHIP, "is_gather_sections: no device scratch (hipMalloc of 4 KB failed)");
    int32_t* d_tot = d_words;          /* [nranks] sections per rank (dst) */
    int32_t* d_go = d_words + nranks;  /* [1] dst's go-ahead for the payload */

    /* ---- phase A: every rank's section total (one int: ncclGather, rccl.h:745) and its per-column counts */
    NCCL_TRY(rccl().Gather(d_offsets + my_cols, d_tot, 1, ncclInt32, dst, c, stream));
    if (equal) {
        if (my_cols > 0)
            NCCL_TRY(rccl().Gather(d_counts, d_all_counts, (size_t)my_cols, ncclInt32, dst, c, stream));
    } else if (int rc = is_gather_i32(comm, dst, cnt.data(), d_counts, d_all_counts, stream_)) {
        return rc;
    }
    int32_t my_total = 0;
    std::vector<int32_t> tot32(nranks, 0);
    HIPG_TRY(hipMemcpyAsync(&my_total, d_offsets + my_cols, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    if (rank == dst)
        HIPG_TRY(hipMemcpyAsync(tot32.data(), d_tot, sizeof(int32_t) * nranks, hipMemcpyDeviceToHost, stream));
    HIPG_TRY(hipStreamSynchronize(stream));

    /* ---- dst decides whether the payload fits and tells every rank (nobody posts a send that would never
     * be received) */
    int32_t go = 1;
    if (rank == dst) {
        int64_t sum = 0;
        for (int r = 0; r < nranks; r++) { h_totals[r] = tot32[r]; sum += tot32[r]; }
        go = ((size_t)sum <= cap_sections && (sum == 0 || d_all_packed != nullptr)) ? 1 : 0;
        HIPG_TRY(hipMemcpyAsync(d_go, &go, sizeof(go), hipMemcpyHostToDevice, stream));
    } else {
        h_totals[rank] = my_total;
    }
    if (nranks > 1) {
        NCCL_TRY(rccl().Broadcast(d_go, d_go, 1, ncclInt32, dst, c, stream));
        HIPG_TRY(hipMemcpyAsync(&go, d_go, sizeof(go), hipMemcpyDeviceToHost, stream));
        HIPG_TRY(hipStreamSynchronize(stream));
    }
    if (!go)
        return isk_fail(IS_ENOMEM, "is_gather_sections: the gathered sections exceed cap_sections on dst "
                                   "(h_totals holds the sizes on dst); nothing was transferred");

    /* ---- phase B: the used sections, 8 int32 each */
    for (int r = 0; r < nranks; r++) cnt[r] = (rank == dst || r == rank) ? 8 * h_totals[r] : 0;
    return is_gather_i32(comm, dst, cnt.data(), (const int32_t*)d_packed, (int32_t*)d_all_packed, stream_);
}

}  // extern "C"
