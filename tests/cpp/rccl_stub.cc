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
//
// Round 6: the same rendezvous between PROCESSES (stub_comm_create_shm): the ranks' staging copies and the barrier live
// in a POSIX shared-memory segment, so that lh_snapshot_merge -- the C-ABI merge itself, not loghisto_amd/merge.py --
// has crossed a process boundary before the first multi-GPU lease (tests/test_gpu_merge_procs.py: two processes, one
// engine each, on the one GPU).
#include <hip/hip_runtime_api.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
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

// The process-shared group: a header and nranks staging areas of `cap` bytes each in one shm segment.
struct ShmHeader {
    std::atomic<uint32_t> ready;      // 0x5a5a once rank 0 has initialised the header
    std::atomic<uint32_t> arrived;
    std::atomic<uint64_t> generation;
    uint32_t nranks;
    uint64_t cap;
    std::atomic<uint64_t> sizes[64];  // bytes each rank staged in the current collective
};
struct ShmGroup {
    ShmHeader *h = nullptr;
    unsigned char *base = nullptr; // first staging area
    size_t map_bytes = 0;
    unsigned char *area(int r) const { return base + (size_t)r * h->cap; }
    // returns false after 120 s: a rank that never arrives must fail the test, not hang it
    bool barrier() const
    {
        const uint64_t gen = h->generation.load(std::memory_order_acquire);
        if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == h->nranks) {
            h->arrived.store(0, std::memory_order_relaxed);
            h->generation.fetch_add(1, std::memory_order_release);
            return true;
        }
        const auto t0 = std::chrono::steady_clock::now();
        while (h->generation.load(std::memory_order_acquire) == gen) {
            sched_yield();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;
        }
        return true;
    }
};

struct Comm {
    Group *group;
    int rank;
    ShmGroup *shm = nullptr; // set: the ranks are processes
    int nranks() const { return shm ? (int)shm->h->nranks : group->nranks; }
};

constexpr int kUint32 = 3, kUint64 = 5, kSum = 0, kMin = 3;

size_t elem_size(int dtype) { return dtype == kUint32 ? 4 : dtype == kUint64 ? 8 : 0; }

// out[i] = reduce over ranks of src[r][first + i]
void reduce(const std::vector<const unsigned char *> &src, int dtype, int op, size_t first, size_t count, void *out)
{
    const int nranks = (int)src.size();
    if (dtype == kUint64) {
        uint64_t *o = static_cast<uint64_t *>(out);
        for (size_t i = 0; i < count; i++) {
            uint64_t acc = op == kMin ? ~0ull : 0ull;
            for (int r = 0; r < nranks; r++) {
                const uint64_t x = reinterpret_cast<const uint64_t *>(src[(size_t)r])[first + i];
                acc = op == kMin ? (x < acc ? x : acc) : acc + x;
            }
            o[i] = acc;
        }
    } else {
        uint32_t *o = static_cast<uint32_t *>(out);
        for (size_t i = 0; i < count; i++) {
            uint32_t acc = op == kMin ? ~0u : 0u;
            for (int r = 0; r < nranks; r++) {
                const uint32_t x = reinterpret_cast<const uint32_t *>(src[(size_t)r])[first + i];
                acc = op == kMin ? (x < acc ? x : acc) : acc + x;
            }
            o[i] = acc;
        }
    }
}

// the same collective between processes: staging areas and barrier in shared memory
int collective_shm(const void *send, void *recv, size_t sendcount, size_t first, size_t recvcount, int dtype, int op,
                   Comm *c, hipStream_t stream, size_t es)
{
    ShmGroup *g = c->shm;
    const size_t bytes = sendcount * es;
    if (bytes > g->h->cap) return 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    if (bytes && hipMemcpy(g->area(c->rank), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    g->h->sizes[c->rank].store(bytes, std::memory_order_release);
    if (!g->barrier()) return 6; // every rank's contribution is staged
    std::vector<const unsigned char *> src;
    for (uint32_t r = 0; r < g->h->nranks; r++) {
        if (g->h->sizes[r].load(std::memory_order_acquire) != bytes) return 5; // ranks disagree on the count
        src.push_back(g->area((int)r));
    }
    std::vector<unsigned char> out(recvcount * es);
    reduce(src, dtype, op, first, recvcount, out.data());
    if (!g->barrier()) return 6; // nobody overwrites its staging area while others still read it
    if (recvcount && hipMemcpy(recv, out.data(), recvcount * es, hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}

int collective(const void *send, void *recv, size_t sendcount, size_t first, size_t recvcount, int dtype, int op,
               Comm *c, hipStream_t stream)
{
    const size_t es = elem_size(dtype);
    if (!es || (op != kSum && op != kMin) || !c) return 4; // ncclInvalidArgument
    if (c->shm) return collective_shm(send, recv, sendcount, first, recvcount, dtype, op, c, stream, es);
    Group *g = c->group;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    std::vector<unsigned char> &mine = g->bufs[(size_t)c->rank];
    mine.resize(sendcount * es);
    if (sendcount && hipMemcpy(mine.data(), send, sendcount * es, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    g->barrier(); // every rank's contribution is staged
    std::vector<unsigned char> out(recvcount * es);
    for (int r = 0; r < g->nranks; r++)
        if (g->bufs[(size_t)r].size() != sendcount * es) return 5; // ranks disagree on the count: a real hang in RCCL
    std::vector<const unsigned char *> src;
    for (int r = 0; r < g->nranks; r++) src.push_back(g->bufs[(size_t)r].data());
    reduce(src, dtype, op, first, recvcount, out.data());
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
    return collective(send, recv, recvcount * (size_t)c->nranks(), recvcount * (size_t)c->rank, recvcount, dtype,
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

// One rank of a group of PROCESSES: every rank calls this with the same name, nranks and cap (bytes of staging per
// rank); rank 0 creates and initialises the segment, the others wait for it (60 s).  *comm is this rank's handle.
int stub_comm_create_shm(const char *name, int nranks, int rank, size_t cap, void **comm)
{
    if (!name || !comm || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return 4;
    const size_t hdr = (sizeof(ShmHeader) + 4095) & ~size_t(4095), bytes = hdr + (size_t)nranks * cap;
    int fd = -1;
    const auto t0 = std::chrono::steady_clock::now();
    if (rank == 0) {
        shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) return 2;
    } else {
        while ((fd = shm_open(name, O_RDWR, 0600)) < 0) {
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return 6;
        }
    }
    void *p = MAP_FAILED;
    for (;;) { // (a rank may open the segment before rank 0 has sized it)
        p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (p != MAP_FAILED) {
            off_t sz = lseek(fd, 0, SEEK_END);
            if (sz >= (off_t)bytes) break;
            munmap(p, bytes);
            p = MAP_FAILED;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { close(fd); return 6; }
    }
    close(fd);
    ShmGroup *g = new ShmGroup;
    g->h = static_cast<ShmHeader *>(p);
    g->base = static_cast<unsigned char *>(p) + hdr;
    g->map_bytes = bytes;
    if (rank == 0) {
        g->h->arrived.store(0);
        g->h->generation.store(0);
        g->h->nranks = (uint32_t)nranks;
        g->h->cap = cap;
        for (auto &s : g->h->sizes) s.store(0);
        g->h->ready.store(0x5a5au, std::memory_order_release);
    } else {
        while (g->h->ready.load(std::memory_order_acquire) != 0x5a5au) {
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return 6;
        }
    }
    *comm = new Comm{nullptr, rank, g};
    return 0;
}

int stub_comm_destroy_shm(const char *name, void *comm)
{
    Comm *c = static_cast<Comm *>(comm);
    if (!c || !c->shm) return 4;
    munmap(c->shm->h, c->shm->map_bytes);
    if (name && c->rank == 0) shm_unlink(name);
    delete c->shm;
    delete c;
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
