// rccl_stub.cc -- TEST INFRASTRUCTURE: an in-process stand-in for librccl.so.
//
// One MI355X is reachable from the test harness, so lh_snapshot_merge (K4, include/loghisto_gpu.h) could only
// ever run with nranks == 1 against the real RCCL (padding to nranks*per rows, owned-row
// offsets and the reduce-scatter receive placement had never seen a second rank).  This library implements the
// two collectives K4 calls -- ncclAllReduce and ncclReduceScatter, for the (uint32, MIN) and (uint64, SUM)
// cases it uses -- as a rendezvous between THREADS of one process: N engines on one GPU play N ranks, each
// merge call runs in its own thread, and the stub reduces their device buffers exactly as RCCL would
// (rank r of a reduce-scatter receives elements [r*recvcount, (r+1)*recvcount) of the element-wise sum).
// Loaded through lh_set_rccl_library(); tests/_stub_merge_driver.py drives it.
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

struct Group {
    int nranks;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<std::vector<unsigned char>> bufs; // one staging copy per rank
    explicit Group(int n) : nranks(n), bufs((size_t)n) {}
    void barrier()
    {
        std::unique_lock<std::mutex> l(mu);
        const uint64_t gen = generation;
        if (++arrived == nranks) {
            arrived = 0;
            generation++;
            cv.notify_all();
        } else {
            cv.wait(l, [&] { return generation != gen; });
        }
    }
};

struct Comm {
    Group *group;
    int rank;
};

constexpr int kUint32 = 3, kUint64 = 5, kSum = 0, kMin = 3;

size_t elem_size(int dtype) { return dtype == kUint32 ? 4 : dtype == kUint64 ? 8 : 0; }

// out[i] = reduce over ranks of bufs[r][first + i]
void reduce(Group *g, int dtype, int op, size_t first, size_t count, void *out)
{
    if (dtype == kUint64) {
        uint64_t *o = static_cast<uint64_t *>(out);
        for (size_t i = 0; i < count; i++) {
            uint64_t acc = op == kMin ? ~0ull : 0ull;
            for (int r = 0; r < g->nranks; r++) {
                const uint64_t x = reinterpret_cast<const uint64_t *>(g->bufs[(size_t)r].data())[first + i];
                acc = op == kMin ? (x < acc ? x : acc) : acc + x;
            }
            o[i] = acc;
        }
    } else {
        uint32_t *o = static_cast<uint32_t *>(out);
        for (size_t i = 0; i < count; i++) {
            uint32_t acc = op == kMin ? ~0u : 0u;
            for (int r = 0; r < g->nranks; r++) {
                const uint32_t x = reinterpret_cast<const uint32_t *>(g->bufs[(size_t)r].data())[first + i];
                acc = op == kMin ? (x < acc ? x : acc) : acc + x;
            }
            o[i] = acc;
        }
    }
}

int collective(const void *send, void *recv, size_t sendcount, size_t first, size_t recvcount, int dtype, int op,
               Comm *c, hipStream_t stream)
{
    const size_t es = elem_size(dtype);
    if (!es || (op != kSum && op != kMin) || !c) return 4; // ncclInvalidArgument
    Group *g = c->group;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    std::vector<unsigned char> &mine = g->bufs[(size_t)c->rank];
    mine.resize(sendcount * es);
    if (sendcount && hipMemcpy(mine.data(), send, sendcount * es, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    g->barrier(); // every rank's contribution is staged
    std::vector<unsigned char> out(recvcount * es);
    for (int r = 0; r < g->nranks; r++)
        if (g->bufs[(size_t)r].size() != sendcount * es) return 5; // ranks disagree on the count: a real hang in RCCL
    reduce(g, dtype, op, first, recvcount, out.data());
    g->barrier(); // nobody overwrites its staging copy while others still read it
    if (recvcount && hipMemcpy(recv, out.data(), recvcount * es, hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}

} // namespace

extern "C" {

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t stream)
{
    return collective(send, recv, count, 0, count, dtype, op, static_cast<Comm *>(comm), stream);
}

int ncclReduceScatter(const void *send, void *recv, size_t recvcount, int dtype, int op, void *comm,
                      hipStream_t stream)
{
    Comm *c = static_cast<Comm *>(comm);
    if (!c) return 4;
    return collective(send, recv, recvcount * (size_t)c->group->nranks, recvcount * (size_t)c->rank, recvcount, dtype,
                      op, c, stream);
}

// test-only helpers: comms[r] is rank r's communicator handle
int stub_comm_create(int nranks, void **comms)
{
    if (nranks < 1 || !comms) return 4;
    Group *g = new Group(nranks);
    for (int r = 0; r < nranks; r++) comms[r] = new Comm{g, r};
    return 0;
}

int stub_comm_destroy(int nranks, void **comms)
{
    if (!comms || nranks < 1) return 4;
    Group *g = static_cast<Comm *>(comms[0])->group;
    for (int r = 0; r < nranks; r++) delete static_cast<Comm *>(comms[r]);
    delete g;
    return 0;
}

} // extern "C"
