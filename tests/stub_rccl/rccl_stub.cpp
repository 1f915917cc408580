// rccl_stub.cpp — TEST INFRASTRUCTURE, never part of the product: a stand-in for librccl.so that lets several processes on ONE GPU
// drive the n_ranks > 1 branch of multiprime_amd/csrc/comm.hip (mp_comm_init, mp_comm_allgatherv's padded slots,
// mp_comm_allgather_i64, mp_eval_candidates_allreduce).  A GPU box of this pool has one device and RCCL refuses two ranks on the
// same device, so the real library can only ever form a world of one there.
//
// It implements exactly the entry points comm.hip resolves with dlsym — ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy,
// ncclAllReduce, ncclAllGather, ncclGetErrorString, and the optional ncclCommCount / ncclCommUserRank (count = the ranks that
// really attached to the segment) — with RCCL's signatures and stream semantics as far as a caller can see them:
// the collective is ordered after the work queued on `stream` before it, and work queued after it sees its result.  Transport:
// a POSIX shared-memory segment named by the unique id (one slot per rank + a sense-reversing barrier), device buffers moved with
// hipMemcpy.  Blocking, chunked, with a time-out instead of a hang.  Selected by MP_RCCL_LIBRARY=<this .so> (comm.hip).
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

extern "C" {

// the slice of rccl.h this stub mirrors (same values as /opt/rocm/include/rccl/rccl.h)
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
struct StubComm;
typedef StubComm *ncclComm_t;

}  // extern "C"

namespace {

constexpr size_t kSlot = 8u << 20;             // bytes of one rank's slot; longer messages go in chunks
constexpr double kTimeoutS = 120.0;

struct Header {
    std::atomic<int> arrived;
    std::atomic<int> sense;
    std::atomic<int> attached;
    std::atomic<int> calls;                    // collectives completed (rank 0 counts): the test reads it back through the id's segment
    int n_ranks;
};

size_t dtype_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclInt32: case ncclUint32: return 4;
        case ncclInt64: case ncclUint64: return 8;
    }
    return 0;
}

}  // namespace

struct StubComm {
    int n_ranks = 0, rank = 0, local_sense = 0;
    char name[64] = {0};
    size_t bytes = 0;
    uint8_t *base = nullptr;
    Header *hdr = nullptr;
    uint8_t *slot(int r) const { return base + 4096 + (size_t)r * kSlot; }
    bool barrier() {
        local_sense ^= 1;
        if (hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n_ranks) {
            hdr->arrived.store(0, std::memory_order_relaxed);
            hdr->sense.store(local_sense, std::memory_order_release);
            return true;
        }
        auto t0 = std::chrono::steady_clock::now();
        while (hdr->sense.load(std::memory_order_acquire) != local_sense) {
            sched_yield();
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) return false;
        }
        return true;
    }
};

// one chunk of a collective: my bytes into my slot, barrier, `consume` reads every slot, barrier
template <typename F>
ncclResult_t exchange(StubComm *c, const uint8_t *send_dev, size_t n, F consume) {
    if (n && hipMemcpy(c->slot(c->rank), send_dev, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!c->barrier()) return ncclSystemError;
    ncclResult_t rc = consume();
    if (!c->barrier()) return ncclSystemError;
    return rc;
}

extern "C" {

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error (rccl stub)";
        case ncclUnhandledCudaError: return "HIP error (rccl stub)";
        case ncclSystemError: return "shared memory / time-out (rccl stub)";
        case ncclInvalidArgument: return "invalid argument (rccl stub)";
        default: return "error (rccl stub)";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id->internal, 0, sizeof id->internal);
    unsigned long long r = 0;
    FILE *f = fopen("/dev/urandom", "rb");
    if (!f || fread(&r, sizeof r, 1, f) != 1) r = (unsigned long long)getpid() * 2654435761ull + (unsigned long long)time(nullptr);
    if (f) fclose(f);
    snprintf(id->internal, 64, "/mp_rccl_stub_%d_%016llx", (int)getpid(), r);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int n_ranks, ncclUniqueId id, int rank) {
    if (!out || n_ranks < 1 || rank < 0 || rank >= n_ranks || id.internal[0] != '/') return ncclInvalidArgument;
    StubComm *c = new StubComm;
    c->n_ranks = n_ranks; c->rank = rank;
    memcpy(c->name, id.internal, 63);
    c->bytes = 4096 + (size_t)n_ranks * kSlot;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->base = static_cast<uint8_t *>(p);
    c->hdr = reinterpret_cast<Header *>(p);                 // a fresh segment is zero-filled: counters start at 0
    c->hdr->n_ranks = n_ranks;
    c->hdr->attached.fetch_add(1);
    auto t0 = std::chrono::steady_clock::now();             // like RCCL, initialisation waits for the whole world
    while (c->hdr->attached.load() < n_ranks) {
        sched_yield();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) {
            munmap(p, c->bytes); shm_unlink(c->name); delete c;
            return ncclSystemError;
        }
    }
    if (!c->barrier()) return ncclSystemError;
    if (rank == 0) shm_unlink(c->name);                      // everyone holds a mapping by now; the name can go
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclInvalidArgument;
    munmap(c->base, c->bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t stream) {
    if (!c || op != ncclSum || dt != ncclInt64 || (count && (!send || !recv))) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    const size_t per = kSlot / 8;
    std::vector<int64_t> acc;
    for (size_t at = 0; at < count || (count == 0 && at == 0); at += per) {
        const size_t n = count ? std::min(per, count - at) : 0;
        acc.assign(n, 0);
        ncclResult_t rc = exchange(c, static_cast<const uint8_t *>(send) + at * 8, n * 8, [&] {
            for (int r = 0; r < c->n_ranks; r++) {
                const int64_t *s = reinterpret_cast<const int64_t *>(c->slot(r));
                for (size_t i = 0; i < n; i++) acc[i] += s[i];
            }
            return ncclSuccess;
        });
        if (rc != ncclSuccess) return rc;
        if (n && hipMemcpy(static_cast<uint8_t *>(recv) + at * 8, acc.data(), n * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        if (count == 0) break;
    }
    if (c->rank == 0) c->hdr->calls.fetch_add(1);
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t dt, ncclComm_t c, hipStream_t stream) {
    const size_t es = dtype_size(dt);
    if (!c || es == 0 || (sendcount && (!send || !recv))) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    const size_t total = sendcount * es;
    // the send buffer may alias the receive buffer's own slot (in-place) or lie elsewhere: it is read before anything is written
    for (size_t at = 0; at < total || (total == 0 && at == 0); at += kSlot) {
        const size_t n = total ? std::min(kSlot, total - at) : 0;
        ncclResult_t rc = exchange(c, static_cast<const uint8_t *>(send) + at, n, [&] {
            for (int r = 0; r < c->n_ranks; r++)
                if (n && hipMemcpy(static_cast<uint8_t *>(recv) + (size_t)r * total + at, c->slot(r), n, hipMemcpyHostToDevice) != hipSuccess)
                    return ncclUnhandledCudaError;
            return ncclSuccess;
        });
        if (rc != ncclSuccess) return rc;
        if (total == 0) break;
    }
    if (c->rank == 0) c->hdr->calls.fetch_add(1);
    return ncclSuccess;
}

// the personalised exchange (an RCCL extension comm.hip prefers to grouped ncclSend / ncclRecv): first every rank's row of counts, then
// n_ranks rounds — in round j rank r's piece for rank (r + j) % n goes through r's slot, in as many chunks as the longest piece of the
// round needs (every rank knows the whole count matrix, so all of them pass the same barriers)
ncclResult_t ncclAllToAllv(const void *send, const size_t sendcounts[], const size_t sdispls[], void *recv, const size_t recvcounts[],
                           const size_t rdispls[], ncclDataType_t dt, ncclComm_t c, hipStream_t stream) {
    const size_t es = dtype_size(dt);
    if (!c || es == 0 || !sendcounts || !sdispls || !recvcounts || !rdispls) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    const int R = c->n_ranks, me = c->rank;
    std::vector<size_t> M((size_t)R * R);
    memcpy(c->slot(me), sendcounts, sizeof(size_t) * (size_t)R);
    if (!c->barrier()) return ncclSystemError;
    for (int r = 0; r < R; r++) memcpy(&M[(size_t)r * R], c->slot(r), sizeof(size_t) * (size_t)R);
    if (!c->barrier()) return ncclSystemError;
    for (int r = 0; r < R; r++)
        if (M[(size_t)r * R + me] != recvcounts[r]) return ncclInvalidArgument;           // what I expect is what they send
    for (int j = 0; j < R; j++) {
        const int dst = (me + j) % R, src = (me - j + R) % R;
        size_t longest = 0;
        for (int r = 0; r < R; r++) longest = std::max(longest, M[(size_t)r * R + (size_t)((r + j) % R)] * es);
        const size_t mine = sendcounts[dst] * es, theirs = recvcounts[src] * es;
        for (size_t at = 0; at < longest; at += kSlot) {
            const size_t n_out = at < mine ? std::min(kSlot, mine - at) : 0, n_in = at < theirs ? std::min(kSlot, theirs - at) : 0;
            ncclResult_t rc = exchange(c, static_cast<const uint8_t *>(send) + sdispls[dst] * es + at, n_out, [&] {
                if (n_in && hipMemcpy(static_cast<uint8_t *>(recv) + rdispls[src] * es + at, c->slot(src), n_in, hipMemcpyHostToDevice) != hipSuccess)
                    return ncclUnhandledCudaError;
                return ncclSuccess;
            });
            if (rc != ncclSuccess) return rc;
        }
    }
    if (me == 0) c->hdr->calls.fetch_add(1);
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int *n) { if (!c || !n) return ncclInvalidArgument; *n = c->hdr->attached.load(); return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int *r) { if (!c || !r) return ncclInvalidArgument; *r = c->rank; return ncclSuccess; }

// test hook (not an RCCL symbol): collectives this communicator has completed, as rank 0 counted them
int mp_rccl_stub_calls(ncclComm_t c) { return c ? c->hdr->calls.load() : -1; }

}  // extern "C"
