// comm.hip — part of libmprime_hip.so: row-shard collectives behind the C ABI of include/mprime.h (mp_comm_*).
// One process per GPU; every rank holds a contiguous block of the alignment's rows and the same windows.  What crosses ranks:
//   * ONE all-reduce (sum, int64) of the [n_candidates x 3] coverage counters per alignment, enqueued on the context's stream
//     straight after the evaluation kernel (no host round trip, no second stream) — mp_comm_allreduce_i64 /
//     mp_eval_candidates_allreduce; the same for the per-window base / pair statistics;
//   * variable-length all-gathers of packed host tables (histogram entries, IUPAC exceptions, per-window results) —
//     mp_comm_allgather_i64 for the lengths, mp_comm_allgatherv for the payloads.
// RCCL over xGMI carries them: messages are KB to a few MB, latency-bound on the point-to-point links, so there is one collective
// per quantity and alignment, never one per window.  librccl is opened on the first mp_comm_* call (dlopen): a single-GPU user
// of the library never loads it.  The communicator's id travels between the ranks by whatever the host has (the Python host
// broadcasts it over its torch.distributed store, INTEGRATION.md shows an MPI / socket host).
#include "common.hpp"

#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>
#include <vector>

using namespace mp;

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    // optional: the personalised exchange in one call (an RCCL extension), else grouped point-to-point calls
    ncclResult_t (*AllToAllv)(const void *, const size_t[], const size_t[], void *, const size_t[], const size_t[], ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;          // optional (mp_comm_describe)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    const char *error = nullptr;
    char path[512] = {0};              // the file the entry points came from (mp_comm_library)
};

// An RCCL that is already mapped into the process (torch's own copy under torch/lib, a host application's) is reused: two copies of
// RCCL in one process would each build their own topology and transports on the same GPU.  dl_iterate_phdr finds it whatever
// name it was loaded under (torch's librccl.so carries no SONAME).
int find_loaded_rccl(struct dl_phdr_info *info, size_t, void *data) {
    const char *n = info->dlpi_name;
    if (!n || !*n) return 0;
    const char *b = strrchr(n, '/');
    b = b ? b + 1 : n;
    if (strncmp(b, "librccl.so", 10) != 0) return 0;
    *static_cast<std::string *>(data) = n;
    return 1;
}

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // MP_RCCL_LIBRARY: an explicit library (the tests' shared-memory stand-in that lets several ranks share one GPU)
        if (const char *forced = getenv("MP_RCCL_LIBRARY")) {
            r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!r.lib) { r.error = "MP_RCCL_LIBRARY could not be opened (dlopen)"; return; }
        }
        if (!r.lib) {
            std::string loaded;
            if (dl_iterate_phdr(find_loaded_rccl, &loaded)) r.lib = dlopen(loaded.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        }
        if (!r.lib)
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.lib) break;
            }
        if (!r.lib) { r.error = "librccl.so not found (dlopen)"; return; }
        auto sym = [&](const char *s) { void *p = dlsym(r.lib, s); if (!p) r.error = "librccl.so lacks an expected symbol"; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.AllToAllv = reinterpret_cast<decltype(r.AllToAllv)>(dlsym(r.lib, "ncclAllToAllv"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.lib, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.lib, "ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.lib, "ncclCommUserRank"));
        Dl_info where;
        if (r.GetUniqueId && dladdr(reinterpret_cast<void *>(r.GetUniqueId), &where) && where.dli_fname)
            snprintf(r.path, sizeof r.path, "%s", where.dli_fname);
    });
    return r;
}

#define NCCLCK(c, call)                                                                                          \
    do {                                                                                                         \
        ncclResult_t r_ = (call);                                                                                \
        if (r_ != ncclSuccess) return fail((c), MP_ERR_DEVICE, "%s: %s", #call, rccl().GetErrorString(r_));     \
    } while (0)

static_assert(MP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the ABI's id is an RCCL unique id");

int need_comm(mp_ctx *c) {
    if (!c) return MP_ERR_ARG;
    if (c->n_ranks <= 0) return fail(c, MP_ERR_ARG, "no communicator (mp_comm_init has not run)");
    return MP_OK;
}

// device scratch of at least n bytes
int scratch(mp_ctx *c, size_t n) {
    if (c->comm_scratch_n >= n) return MP_OK;
    dev_free(c, &c->comm_scratch, c->comm_scratch_n);
    c->comm_scratch_n = 0;
    int rc = dev_alloc(c, &c->comm_scratch, n);
    if (rc) return rc;
    c->comm_scratch_n = n;
    return MP_OK;
}

}  // namespace

namespace mp {

void free_comm(mp_ctx *c) {
    if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(reinterpret_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
    c->n_ranks = 0; c->rank = 0;
    dev_free(c, &c->comm_scratch, c->comm_scratch_n);
    c->comm_scratch_n = 0;
}

}  // namespace mp

extern "C" {

int mp_comm_unique_id(uint8_t *id) {
    if (!id) return MP_ERR_ARG;
    Rccl &r = rccl();
    if (r.error) return MP_ERR_DEVICE;
    ncclUniqueId u;
    if (r.GetUniqueId(&u) != ncclSuccess) return MP_ERR_DEVICE;
    memcpy(id, u.internal, MP_COMM_ID_BYTES);
    return MP_OK;
}

int mp_comm_init(mp_ctx *c, int32_t n_ranks, int32_t rank, const uint8_t *id) {
    if (!c) return MP_ERR_ARG;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks || (n_ranks > 1 && !id)) return fail(c, MP_ERR_ARG, "mp_comm_init: bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    free_comm(c);
    // a world of one needs no RCCL: every collective is the identity.  MP_COMM_FORCE_RCCL=1 (tests on a single GPU) creates the
    // one-rank communicator anyway and sends every collective through RCCL
    if (n_ranks > 1 || (getenv("MP_COMM_FORCE_RCCL") && id)) {
        Rccl &r = rccl();
        if (r.error) return fail(c, MP_ERR_DEVICE, "mp_comm_init: %s", r.error);
        ncclUniqueId u;
        memcpy(u.internal, id, MP_COMM_ID_BYTES);
        ncclComm_t comm = nullptr;
        NCCLCK(c, r.CommInitRank(&comm, n_ranks, u, rank));
        c->comm = comm;
    }
    c->n_ranks = n_ranks; c->rank = rank;
    return MP_OK;
}

int mp_comm_destroy(mp_ctx *c) {
    if (!c) return MP_ERR_ARG;
    (void)hipSetDevice(c->dev);
    (void)hipStreamSynchronize(c->stream);
    free_comm(c);
    return MP_OK;
}

int mp_comm_describe(mp_ctx *c, int32_t *ranks_seen, char *library_path, int32_t path_bytes) {
    int rc = need_comm(c);
    if (rc) return rc;
    if (!ranks_seen || (path_bytes > 0 && !library_path)) return fail(c, MP_ERR_ARG, "mp_comm_describe: null output");
    ranks_seen[0] = c->n_ranks; ranks_seen[1] = c->rank;
    if (path_bytes > 0) library_path[0] = 0;
    if (!c->comm) return MP_OK;
    Rccl &r = rccl();
    int n = -1, me = -1;
    if (r.CommCount && r.CommUserRank) {
        NCCLCK(c, r.CommCount(reinterpret_cast<ncclComm_t>(c->comm), &n));
        NCCLCK(c, r.CommUserRank(reinterpret_cast<ncclComm_t>(c->comm), &me));
        ranks_seen[0] = n; ranks_seen[1] = me;
    }
    if (path_bytes > 0) snprintf(library_path, (size_t)path_bytes, "%s", r.path);
    return MP_OK;
}

int mp_comm_allreduce_i64(mp_ctx *c, int64_t *device_buf, int64_t n) {
    int rc = need_comm(c);
    if (rc) return rc;
    if (n < 0 || (n && !device_buf)) return fail(c, MP_ERR_ARG, "mp_comm_allreduce_i64: bad arguments");
    if (!c->comm || n == 0) return MP_OK;
    HIPCK(c, hipSetDevice(c->dev));
    NCCLCK(c, rccl().AllReduce(device_buf, device_buf, (size_t)n, ncclInt64, ncclSum, reinterpret_cast<ncclComm_t>(c->comm), c->stream));
    return MP_OK;
}

int mp_comm_allreduce_host_i64(mp_ctx *c, int64_t *host_buf, int64_t n) {
    int rc = need_comm(c);
    if (rc) return rc;
    if (n < 0 || (n && !host_buf)) return fail(c, MP_ERR_ARG, "mp_comm_allreduce_host_i64: bad arguments");
    if (!c->comm || n == 0) return MP_OK;
    HIPCK(c, hipSetDevice(c->dev));
    if ((rc = scratch(c, (size_t)n * 8))) return rc;
    HIPCK(c, hipMemcpyAsync(c->comm_scratch, host_buf, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    if ((rc = mp_comm_allreduce_i64(c, reinterpret_cast<int64_t *>(c->comm_scratch), n))) return rc;
    HIPCK(c, hipMemcpyAsync(host_buf, c->comm_scratch, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MP_OK;
}

int mp_comm_allgather_i64(mp_ctx *c, int64_t value, int64_t *out) {
    int rc = need_comm(c);
    if (rc) return rc;
    if (!out) return fail(c, MP_ERR_ARG, "mp_comm_allgather_i64: null output");
    if (!c->comm) { out[0] = value; return MP_OK; }
    HIPCK(c, hipSetDevice(c->dev));
    const size_t R = (size_t)c->n_ranks;
    if ((rc = scratch(c, (R + 1) * 8))) return rc;
    int64_t *d = reinterpret_cast<int64_t *>(c->comm_scratch);
    HIPCK(c, hipMemcpyAsync(d + R, &value, 8, hipMemcpyHostToDevice, c->stream));
    NCCLCK(c, rccl().AllGather(d + R, d, 1, ncclInt64, reinterpret_cast<ncclComm_t>(c->comm), c->stream));
    HIPCK(c, hipMemcpyAsync(out, d, R * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MP_OK;
}

int mp_comm_allgatherv(mp_ctx *c, const void *send, int64_t n_bytes, const int64_t *counts, void *recv) {
    int rc = need_comm(c);
    if (rc) return rc;
    if (n_bytes < 0 || !counts || (n_bytes && !send)) return fail(c, MP_ERR_ARG, "mp_comm_allgatherv: bad arguments");
    if (counts[c->rank] != n_bytes) return fail(c, MP_ERR_ARG, "mp_comm_allgatherv: counts[rank] differs from the bytes sent");
    int64_t total = 0, widest = 0;
    for (int r = 0; r < c->n_ranks; r++) {
        if (counts[r] < 0) return fail(c, MP_ERR_ARG, "mp_comm_allgatherv: negative count");
        total += counts[r];
        widest = std::max(widest, counts[r]);
    }
    if (total && !recv) return fail(c, MP_ERR_ARG, "mp_comm_allgatherv: null output");
    if (!c->comm) { if (n_bytes) memcpy(recv, send, (size_t)n_bytes); return MP_OK; }
    if (widest == 0) return MP_OK;
    HIPCK(c, hipSetDevice(c->dev));
    // every rank's payload padded to the widest one (16-byte granules), gathered in one collective, unpadded on the way out
    const size_t slot = ((size_t)widest + 15) / 16 * 16, R = (size_t)c->n_ranks;
    if ((rc = scratch(c, slot * (R + 1)))) return rc;
    uint8_t *d = c->comm_scratch;
    if (n_bytes) HIPCK(c, hipMemcpyAsync(d + slot * R, send, (size_t)n_bytes, hipMemcpyHostToDevice, c->stream));
    NCCLCK(c, rccl().AllGather(d + slot * R, d, slot, ncclUint8, reinterpret_cast<ncclComm_t>(c->comm), c->stream));
    size_t at = 0;
    for (size_t r = 0; r < R; r++) {
        if (counts[r]) HIPCK(c, hipMemcpyAsync(static_cast<uint8_t *>(recv) + at, d + slot * r, (size_t)counts[r], hipMemcpyDeviceToHost, c->stream));
        at += (size_t)counts[r];
    }
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MP_OK;
}

int mp_comm_alltoall_counts(mp_ctx *c, const int64_t *send_counts, int64_t *recv_counts) {
    int rc = need_comm(c);
    if (rc) return rc;
    if (!send_counts || !recv_counts) return fail(c, MP_ERR_ARG, "mp_comm_alltoall_counts: null argument");
    for (int r = 0; r < c->n_ranks; r++)
        if (send_counts[r] < 0) return fail(c, MP_ERR_ARG, "mp_comm_alltoall_counts: negative count");
    if (!c->comm) { recv_counts[0] = send_counts[0]; return MP_OK; }
    HIPCK(c, hipSetDevice(c->dev));
    // every rank's row of counts to everyone (R x R int64: bytes), this rank's column is what it receives
    const size_t R = (size_t)c->n_ranks;
    if ((rc = scratch(c, (R * R + R) * 8))) return rc;
    int64_t *d = reinterpret_cast<int64_t *>(c->comm_scratch);
    HIPCK(c, hipMemcpyAsync(d + R * R, send_counts, R * 8, hipMemcpyHostToDevice, c->stream));
    NCCLCK(c, rccl().AllGather(d + R * R, d, R, ncclInt64, reinterpret_cast<ncclComm_t>(c->comm), c->stream));
    std::vector<int64_t> all(R * R);
    HIPCK(c, hipMemcpyAsync(all.data(), d, R * R * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    for (size_t r = 0; r < R; r++) recv_counts[r] = all[r * R + (size_t)c->rank];
    return MP_OK;
}

int mp_comm_alltoallv(mp_ctx *c, const void *send, const int64_t *send_counts, void *recv, const int64_t *recv_counts) {
    int rc = need_comm(c);
    if (rc) return rc;
    if (!send_counts || !recv_counts) return fail(c, MP_ERR_ARG, "mp_comm_alltoallv: null counts");
    const size_t R = (size_t)c->n_ranks;
    // 16-byte granules per piece on the device (the pieces of one rank are laid out one after the other, padded)
    std::vector<size_t> s_cnt(R), s_dis(R), r_cnt(R), r_dis(R);
    size_t s_tot = 0, r_tot = 0, s_bytes = 0, r_bytes = 0;
    for (size_t r = 0; r < R; r++) {
        if (send_counts[r] < 0 || recv_counts[r] < 0) return fail(c, MP_ERR_ARG, "mp_comm_alltoallv: negative count");
        s_cnt[r] = (size_t)send_counts[r]; s_dis[r] = s_tot; s_tot += (s_cnt[r] + 15) / 16 * 16; s_bytes += s_cnt[r];
        r_cnt[r] = (size_t)recv_counts[r]; r_dis[r] = r_tot; r_tot += (r_cnt[r] + 15) / 16 * 16; r_bytes += r_cnt[r];
    }
    if ((s_bytes && !send) || (r_bytes && !recv)) return fail(c, MP_ERR_ARG, "mp_comm_alltoallv: null buffer");
    if (!c->comm) {
        if (recv_counts[0] != send_counts[0]) return fail(c, MP_ERR_ARG, "mp_comm_alltoallv: a world of one receives what it sends");
        if (s_bytes) memcpy(recv, send, s_bytes);
        return MP_OK;
    }
    HIPCK(c, hipSetDevice(c->dev));
    Rccl &L = rccl();
    if (!L.AllToAllv && !(L.Send && L.Recv && L.GroupStart && L.GroupEnd)) return fail(c, MP_ERR_DEVICE, "mp_comm_alltoallv: this librccl has neither ncclAllToAllv nor ncclSend / ncclRecv");
    if ((rc = scratch(c, s_tot + r_tot + 32))) return rc;
    uint8_t *ds = c->comm_scratch, *dr = c->comm_scratch + (s_tot + 15) / 16 * 16;
    {
        size_t at = 0;
        for (size_t r = 0; r < R; r++) {
            if (s_cnt[r]) HIPCK(c, hipMemcpyAsync(ds + s_dis[r], static_cast<const uint8_t *>(send) + at, s_cnt[r], hipMemcpyHostToDevice, c->stream));
            at += s_cnt[r];
        }
    }
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(c->comm);
    if (L.AllToAllv) {
        NCCLCK(c, L.AllToAllv(ds, s_cnt.data(), s_dis.data(), dr, r_cnt.data(), r_dis.data(), ncclUint8, comm, c->stream));
    } else {
        NCCLCK(c, L.GroupStart());
        for (size_t r = 0; r < R; r++) {
            if (s_cnt[r]) NCCLCK(c, L.Send(ds + s_dis[r], s_cnt[r], ncclUint8, (int)r, comm, c->stream));
            if (r_cnt[r]) NCCLCK(c, L.Recv(dr + r_dis[r], r_cnt[r], ncclUint8, (int)r, comm, c->stream));
        }
        NCCLCK(c, L.GroupEnd());
    }
    {
        size_t at = 0;
        for (size_t r = 0; r < R; r++) {
            if (r_cnt[r]) HIPCK(c, hipMemcpyAsync(static_cast<uint8_t *>(recv) + at, dr + r_dis[r], r_cnt[r], hipMemcpyDeviceToHost, c->stream));
            at += r_cnt[r];
        }
    }
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MP_OK;
}

int mp_eval_candidates_allreduce(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint64_t sF, uint64_t sR, int64_t *out) {
    int rc = need_comm(c);
    if (rc) return rc;
    if ((rc = mp_eval_upload(c, n_cand, cw, codes, sF, sR))) return rc;
    if (n_cand == 0) return MP_OK;
    if (!out) return fail(c, MP_ERR_ARG, "null output");
    if (c->tmp_out_n < 3 * n_cand) {
        dev_free(c, &c->tmp_out, (size_t)c->tmp_out_n);
        c->tmp_out_n = 0;
        if ((rc = dev_alloc(c, &c->tmp_out, (size_t)3 * n_cand))) return rc;
        c->tmp_out_n = 3 * n_cand;
    }
    // kernel, collective and copy queue up on one stream: nothing waits on the host in between
    if ((rc = mp_eval_launch(c, (int64_t *)c->tmp_out))) return rc;
    if ((rc = mp_comm_allreduce_i64(c, (int64_t *)c->tmp_out, (int64_t)3 * n_cand))) return rc;
    HIPCK(c, hipMemcpyAsync(out, c->tmp_out, sizeof(int64_t) * 3 * (size_t)n_cand, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MP_OK;
}

}  // extern "C"
