// Native data-parallel collective: the CTC-loss gradient all-reduce of the fine-tune step (reference: tf.distribute's cross-replica
// SUM behind src/main.py:148-156,192,198-200) issued by the library itself over RCCL, for hosts that do not carry torch.distributed
// (INTEGRATION.md section 3's torch-free stub) and as an alternative engine for wav2vec2.Trainer (collective="native").
//
// One communicator per model and process (one process per GPU).  RCCL is bound at run time (dlopen of librccl.so.1 -- the copy a
// host process already loaded, e.g. PyTorch's, is reused: one RCCL per process -- and dlsym of the six entry points), so the library
// keeps loading on hosts without RCCL; only w2v2_comm_* then fail, loudly.
//
// Protocol per step (mirrors wav2vec2/training.py::Trainer.all_reduce_gradients):
//   w2v2_train_backward(m, dlogits, s)          enqueues the backward; records one event per gradient bucket (lm_head, layers N-1 .. 0, front)
//   for k in buckets: w2v2_allreduce_bucket(m, k, algo)
//                                               the library's communication stream waits for bucket k's event ONLY, then reduces the
//                                               bucket's trainable runs in place: the upper layers' collectives run under the backward of
//                                               the lower ones.  algo 0: ncclAllReduce; algo 1: ncclReduceScatter + ncclAllGather over the
//                                               run's world-divisible body (every rank owns 1 / world of it) + an all-reduce of the tail
//   w2v2_allreduce_finish(m, s)                 `s` (the optimizer's stream) waits for the communication stream
// xGMI is point to point (7 links x ~153 GB/s per GPU): both forms move 2 (N - 1) / N x payload per GPU; which ring / tree RCCL builds
// for them is its choice -- the two algos exist so that the 8-GPU run can time both (bench.py --collective native[-rs]).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>
#include <utility>
#include <vector>

#include "model.h"

struct Comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;       // the communication stream (non-blocking, highest priority: collectives must not queue behind compute)
    hipEvent_t done = nullptr;
    int rank = 0, world = 1;
    int64_t bytes_last_step = 0;        // payload enqueued since the last w2v2_allreduce_finish
};

static_assert(sizeof(ncclUniqueId) == W2V2_COMM_ID_BYTES, "include/w2v2.h: W2V2_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // RTLD_NOLOAD first: a librccl.so.1 the process already holds (PyTorch's) is THE RCCL of this process
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.handle) break;
            }
        if (!r.handle) return;
        auto sym = [&](const char* n) { return dlsym(r.handle, n); };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.ReduceScatter = reinterpret_cast<decltype(r.ReduceScatter)>(sym("ncclReduceScatter"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.ReduceScatter && r.AllGather && r.GetVersion &&
               r.GetErrorString && r.GroupStart && r.GroupEnd;
    });
    return r;
}

int need_rccl() {
    if (!rccl().ok) {
        w2v2::set_error("w2v2_comm: RCCL is not available in this process (dlopen librccl.so.1: %s)", rccl().handle ? "missing symbols" : dlerror());
        return W2V2_ESTATE;
    }
    return W2V2_OK;
}

#define W2V2_NCCL_CHECK(expr)                                                                   \
    do {                                                                                        \
        const ncclResult_t r_ = (expr);                                                         \
        if (r_ != ncclSuccess) {                                                                \
            w2v2::set_error("%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(r_), __FILE__, __LINE__); \
            return W2V2_EHIP;                                                                   \
        }                                                                                       \
    } while (0)

}  // namespace

void w2v2_comm_free(w2v2_model* m) {
    if (!m || !m->comm) return;
    Comm* c = m->comm;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    m->comm = nullptr;
}

extern "C" {

int w2v2_comm_unique_id(uint8_t* id_out, int32_t nbytes) {
    W2V2_REQUIRE(id_out && nbytes == (int32_t)sizeof(ncclUniqueId), "comm_unique_id: the id buffer must be %d bytes", (int)sizeof(ncclUniqueId));
    if (int e = need_rccl()) return e;
    ncclUniqueId id;
    W2V2_NCCL_CHECK(rccl().GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return W2V2_OK;
}

int w2v2_comm_init(w2v2_model* m, const uint8_t* unique_id, int32_t nbytes, int32_t rank, int32_t world) {
    W2V2_REQUIRE(m && unique_id && nbytes == (int32_t)sizeof(ncclUniqueId), "comm_init: null argument / the id must be %d bytes", (int)sizeof(ncclUniqueId));
    W2V2_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d outside a world of %d", rank, world);
    if (int e = need_rccl()) return e;
    w2v2_comm_free(m);
    Comm* c = new Comm();
    c->rank = rank;
    c->world = world;
    m->comm = c;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    W2V2_NCCL_CHECK(rccl().CommInitRank(&c->comm, world, id, rank));      // (the communicator binds the CURRENT device: one process per GPU)
    int lo = 0, hi = 0;
    W2V2_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    W2V2_HIP_CHECK(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
    W2V2_HIP_CHECK(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    return W2V2_OK;
}

int w2v2_comm_info(const w2v2_model* m, int32_t* rank, int32_t* world, int32_t* rccl_version) {
    W2V2_REQUIRE(m, "comm_info: null model");
    if (rank) *rank = m->comm ? m->comm->rank : 0;
    if (world) *world = m->comm ? m->comm->world : 0;      // 0: no communicator
    if (rccl_version) {
        *rccl_version = 0;
        if (rccl().ok) {
            int v = 0;
            if (rccl().GetVersion(&v) == ncclSuccess) *rccl_version = v;
        }
    }
    return W2V2_OK;
}

int w2v2_comm_destroy(w2v2_model* m) {
    W2V2_REQUIRE(m, "comm_destroy: null model");
    w2v2_comm_free(m);
    return W2V2_OK;
}

int w2v2_allreduce_num_runs(w2v2_model* m, int32_t k, int32_t* count) {
    W2V2_REQUIRE(m && count, "allreduce_num_runs: null argument");
    std::vector<std::pair<int64_t, int64_t>> runs;
    if (int e = w2v2_train_trainable_runs(m, k, &runs, nullptr)) return e;
    *count = (int32_t)runs.size();
    return W2V2_OK;
}

int w2v2_allreduce_run(w2v2_model* m, int32_t k, int32_t i, int64_t* offset, int64_t* numel) {
    W2V2_REQUIRE(m && offset && numel, "allreduce_run: null argument");
    std::vector<std::pair<int64_t, int64_t>> runs;
    if (int e = w2v2_train_trainable_runs(m, k, &runs, nullptr)) return e;
    W2V2_REQUIRE(i >= 0 && i < (int32_t)runs.size(), "allreduce_run: run %d outside [0, %d) of bucket %d", i, (int)runs.size(), k);
    *offset = runs[i].first;
    *numel = runs[i].second;
    return W2V2_OK;
}

int w2v2_allreduce_bucket(w2v2_model* m, int32_t k, int32_t algo) {
    W2V2_REQUIRE(m && m->comm && m->comm->comm, "allreduce_bucket: no communicator (w2v2_comm_init first)");
    W2V2_REQUIRE(algo == 0 || algo == 1, "allreduce_bucket: algo %d (0 = all-reduce, 1 = reduce-scatter + all-gather)", algo);
    Comm* c = m->comm;
    std::vector<std::pair<int64_t, int64_t>> runs;
    float* grads = nullptr;
    if (int e = w2v2_train_trainable_runs(m, k, &runs, &grads)) return e;
    if (runs.empty()) return W2V2_OK;
    if (int e = w2v2_train_bucket_wait(m, k, c->stream)) return e;      // this bucket's slice of the flat buffer is final; nothing else is waited for
    const size_t W = (size_t)c->world;
    W2V2_NCCL_CHECK(rccl().GroupStart());      // one launch for the bucket's runs (a layer: one run; the front: two around the frozen conv stack)
    for (const auto& r : runs) {
        float* p = grads + r.first;
        const size_t n = (size_t)r.second;
        c->bytes_last_step += (int64_t)n * 4;
        if (algo == 0 || W == 1 || n < W * 1024) {
            W2V2_NCCL_CHECK(rccl().AllReduce(p, p, n, ncclFloat32, ncclSum, c->comm, c->stream));
            continue;
        }
        // rank r owns elements [r chunk, (r + 1) chunk) of the body: reduce-scatter into its own chunk (in place: recv = send + rank chunk)
        const size_t chunk = n / W, body = chunk * W;
        W2V2_NCCL_CHECK(rccl().ReduceScatter(p, p + (size_t)c->rank * chunk, chunk, ncclFloat32, ncclSum, c->comm, c->stream));
        if (body < n) W2V2_NCCL_CHECK(rccl().AllReduce(p + body, p + body, n - body, ncclFloat32, ncclSum, c->comm, c->stream));
    }
    W2V2_NCCL_CHECK(rccl().GroupEnd());
    if (algo == 1 && W > 1) {                  // (a second group: the gather reads what the scatter of the same run wrote)
        W2V2_NCCL_CHECK(rccl().GroupStart());
        for (const auto& r : runs) {
            float* p = grads + r.first;
            const size_t n = (size_t)r.second;
            if (n < W * 1024) continue;
            const size_t chunk = n / W;
            W2V2_NCCL_CHECK(rccl().AllGather(p + (size_t)c->rank * chunk, p, chunk, ncclFloat32, c->comm, c->stream));
        }
        W2V2_NCCL_CHECK(rccl().GroupEnd());
    }
    return W2V2_OK;
}

int w2v2_allreduce_finish(w2v2_model* m, void* stream, int64_t* payload_bytes) {
    W2V2_REQUIRE(m && m->comm && m->comm->stream, "allreduce_finish: no communicator (w2v2_comm_init first)");
    Comm* c = m->comm;
    W2V2_HIP_CHECK(hipEventRecord(c->done, c->stream));
    W2V2_HIP_CHECK(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), c->done, 0));
    if (payload_bytes) *payload_bytes = c->bytes_last_step;
    c->bytes_last_step = 0;
    return W2V2_OK;
}

}  // extern "C"
