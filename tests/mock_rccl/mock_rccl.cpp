/*
 * mock_rccl.cpp -- TEST INFRASTRUCTURE, never part of the product: the twelve RCCL entry points csrc/is_gather.hip
 * resolves with dlsym, implemented over POSIX shared memory + host staging, so that the C-ABI gather
 * (is_gather_i32, is_gather_sections, Stixels::ComputeBatchGather) can run with nranks > 1 on a box that has ONE
 * GPU: real RCCL refuses two ranks on one device, and the GPU pool leases single-GPU boxes.  What this exercises
 * is OUR side of the exchange -- the order of the collectives on every rank, the counts and offsets, dst's
 * go-ahead, the uneven-shard path -- with the semantics of the real calls (rccl.h:678-745): sends are buffered
 * (never wait for the receiver), a receive blocks until its message is there, a collective is ordered behind the
 * work already queued on its stream.  Selected with IS_RCCL_LIB=<path of this library> (is_gather.hip).
 *
 *   hipcc -shared -fPIC -O1 tests/mock_rccl/mock_rccl.cpp -lrt -o tests/mock_rccl/libmock_rccl.so
 */
#include <hip/hip_runtime_api.h>

#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <vector>

namespace {
constexpr int MAXR = 4;
constexpr size_t CAP = (size_t)8 << 20; /* bytes of one (src -> dst) channel (the object is sparse: only touched pages exist) */
struct Channel {
    std::atomic<uint64_t> head; /* bytes written (producer) */
    std::atomic<uint64_t> tail; /* bytes consumed (consumer) */
    char data[CAP];
};
struct Shm {
    std::atomic<int> arrived, left;
    Channel ch[MAXR][MAXR];
};
struct Comm {
    Shm* shm;
    int rank, nranks;
    char name[64];
};
double now() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}
constexpr double TIMEOUT_S = 120.0;
size_t type_bytes(int t) { return (t == 0 || t == 1) ? 1 : (t == 4 || t == 5 || t == 8) ? 8 : (t == 6 || t == 9) ? 2 : 4; }

/* copy in / out of the byte ring (wraps) */
void ring_write(Channel& c, uint64_t at, const void* src, size_t n) {
    const size_t o = at % CAP, first = n < CAP - o ? n : CAP - o;
    memcpy(c.data + o, src, first);
    if (first < n) memcpy(c.data, (const char*)src + first, n - first);
}
void ring_read(Channel& c, uint64_t at, void* dst, size_t n) {
    const size_t o = at % CAP, first = n < CAP - o ? n : CAP - o;
    memcpy(dst, c.data + o, first);
    if (first < n) memcpy((char*)dst + first, c.data, n - first);
}
int send_bytes(Comm* c, int peer, const void* d_src, size_t n, hipStream_t stream) {
    if (n + 8 > CAP) return 5;
    std::vector<char> tmp(n);
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    if (n && hipMemcpy(tmp.data(), d_src, n, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    Channel& ch = c->shm->ch[c->rank][peer];
    const double t0 = now();
    uint64_t h = ch.head.load(std::memory_order_relaxed);
    while (h + 8 + n - ch.tail.load(std::memory_order_acquire) > CAP) /* (full: the receiver is behind) */
        if (now() - t0 > TIMEOUT_S) return 6;
    const uint64_t len = n;
    ring_write(ch, h, &len, 8);
    ring_write(ch, h + 8, tmp.data(), n);
    ch.head.store(h + 8 + n, std::memory_order_release);
    return 0;
}
int recv_bytes(Comm* c, int peer, void* d_dst, size_t n, hipStream_t stream) {
    Channel& ch = c->shm->ch[peer][c->rank];
    const double t0 = now();
    const uint64_t t = ch.tail.load(std::memory_order_relaxed);
    while (ch.head.load(std::memory_order_acquire) - t < 8)
        if (now() - t0 > TIMEOUT_S) return 6;
    uint64_t len = 0;
    ring_read(ch, t, &len, 8);
    if (len != n) { /* the two sides disagree about a size: exactly what this mock is here to catch */
        fprintf(stderr, "mock_rccl: rank %d expects %zu bytes from rank %d, which sent %llu\n", c->rank, n, peer,
                (unsigned long long)len);
        return 4;
    }
    while (ch.head.load(std::memory_order_acquire) - t < 8 + n)
        if (now() - t0 > TIMEOUT_S) return 6;
    std::vector<char> tmp(n);
    ring_read(ch, t + 8, tmp.data(), n);
    ch.tail.store(t + 8 + n, std::memory_order_release);
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    if (n && hipMemcpy(d_dst, tmp.data(), n, hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    unsigned char r[8] = {0};
    FILE* f = fopen("/dev/urandom", "rb");
    if (f) { if (fread(r, 1, 8, f) != 8) r[0] = 1; fclose(f); }
    snprintf(id->internal, sizeof(id->internal), "/is_mock_rccl_%d_%02x%02x%02x%02x%02x%02x%02x%02x", (int)getpid(), r[0],
             r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
    return 0;
}
int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return 4;
    Comm* c = new Comm;
    c->rank = rank; c->nranks = nranks;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return 2;
    if (ftruncate(fd, sizeof(Shm)) != 0) return 2;
    void* p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); /* (a fresh object reads as zeros) */
    close(fd);
    if (p == MAP_FAILED) return 2;
    c->shm = (Shm*)p;
    c->shm->arrived.fetch_add(1);
    const double t0 = now();
    while (c->shm->arrived.load() < nranks)
        if (now() - t0 > TIMEOUT_S) return 6;
    *comm = c;
    return 0;
}
int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    const bool last = c->shm->left.fetch_add(1) + 1 == c->nranks;
    munmap(c->shm, sizeof(Shm));
    if (last) shm_unlink(c->name);
    delete c;
    return 0;
}
int ncclCommCount(const void* comm, int* n) { *n = ((const Comm*)comm)->nranks; return 0; }
int ncclCommUserRank(const void* comm, int* r) { *r = ((const Comm*)comm)->rank; return 0; }
int ncclSend(const void* buf, size_t count, int type, int peer, void* comm, hipStream_t s) {
    return send_bytes((Comm*)comm, peer, buf, count * type_bytes(type), s);
}
int ncclRecv(void* buf, size_t count, int type, int peer, void* comm, hipStream_t s) {
    return recv_bytes((Comm*)comm, peer, buf, count * type_bytes(type), s);
}
int ncclGather(const void* send, void* recv, size_t count, int type, int root, void* comm, hipStream_t s) {
    Comm* c = (Comm*)comm;
    const size_t n = count * type_bytes(type);
    if (c->rank != root) return send_bytes(c, root, send, n, s);
    for (int r = 0; r < c->nranks; r++) {
        char* dst = (char*)recv + (size_t)r * n;
        if (r == root) {
            if (hipStreamSynchronize(s) != hipSuccess) return 1;
            if (n && dst != send && hipMemcpy(dst, send, n, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
        } else if (int e = recv_bytes(c, r, dst, n, s)) {
            return e;
        }
    }
    return 0;
}
int ncclBroadcast(const void* send, void* recv, size_t count, int type, int root, void* comm, hipStream_t s) {
    Comm* c = (Comm*)comm;
    const size_t n = count * type_bytes(type);
    if (c->rank != root) return recv_bytes(c, root, recv, n, s);
    for (int r = 0; r < c->nranks; r++)
        if (r != root)
            if (int e = send_bytes(c, r, send, n, s)) return e;
    if (recv != send && n && hipMemcpy(recv, send, n, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
    return 0;
}
int ncclGroupStart() { return 0; } /* (the grouped calls of is_gather_i32 are receives on one rank: in order is fine) */
int ncclGroupEnd() { return 0; }
const char* ncclGetErrorString(int e) {
    switch (e) {
        case 0: return "no error";
        case 1: return "mock: HIP call failed";
        case 2: return "mock: shared memory";
        case 4: return "mock: invalid argument / size mismatch between ranks";
        case 5: return "mock: message larger than a channel";
        case 6: return "mock: timeout (a rank never posted the matching call: the collective sequence diverged)";
        default: return "mock: error";
    }
}
}
