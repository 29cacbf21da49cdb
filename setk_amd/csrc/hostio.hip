// hostio.hip -- the READ stage of the command line's host pipeline (setk_amd/pipeline.py) as one
// call per batch: every payload of a batch (a wave file's 16-bit frames, a mask's float32 rows,
// exactly as stored) is copied from the page cache into the batch's page-locked slab by a pool
// of native threads.
//
// Replaces (funcwj/setk): the per-utterance read of libs/data_handler.py:345-413 (WaveReader /
// ScriptReader._load: open, decode, one utterance at a time) under
// apply_adaptive_beamformer.py:130-178.  Host code only; nothing here touches the device.
//
// Why native: the interpreter's readers need ~10 acquisitions of the interpreter lock per payload
// (open, fstat, mmap, madvise, view, copy, close ...), each of which can wait a switch interval
// while the planning thread runs Python -- the same mmap + copy reaches 40 - 52 GB/s alone and
// 11 - 19 GB/s inside the pipeline (profiles/round5_read_small.txt, round5_e2e_p1_sweep.txt).
// How a payload is read is unchanged: a MAP_SHARED mapping advised MADV_SEQUENTIAL and one memcpy
// out of it (a read() marks every page accessed and moves fresh pages between the kernel's LRU
// lists under a shared lock: 15 - 19 GB/s however many threads), preadv for small payloads (a
// map / unmap pair costs a TLB shootdown).
//
// A mapped file that another process TRUNCATES between the fstat and the copy raises SIGBUS in
// the copying thread (the pread form reports EIO for the same race).  Inputs that can shrink
// while they are read (network or scratch file systems with concurrent writers) should run with
// mmap_min_bytes above every payload (command line: SETK_READ_MODE=preadv).
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/setk_hip.h"

namespace {

struct ReadBatch {
    int n;
    const char* const* paths;
    const long long* offsets;
    const long long* nbytes;
    void* const* dst;
    long long mmap_min;
    int* status;
    int next = 0;  // guarded by Pool::m
    std::atomic<int> left;
    std::mutex m;
    std::condition_variable cv;
    bool done = false;
};

// 0, or the errno of the failing call; EIO: the file ends inside the payload
int read_one(const char* path, long long off, long long nb, void* dst, long long mmap_min) {
    if (nb <= 0) return 0;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return errno ? errno : EIO;
    int rc = 0;
    bool copied = false;
    if (nb >= mmap_min) {
        struct stat st;
        if (fstat(fd, &st) == 0 && (long long)st.st_size >= off + nb) {
            static const long long page = sysconf(_SC_PAGESIZE);
            const long long lo = off & ~(page - 1);
            const size_t len = (size_t)(off - lo + nb);
            void* m = mmap(nullptr, len, PROT_READ, MAP_SHARED, fd, (off_t)lo);
            if (m != MAP_FAILED) {
                madvise(m, len, MADV_SEQUENTIAL);
                memcpy(dst, static_cast<const char*>(m) + (off - lo), (size_t)nb);
                munmap(m, len);
                copied = true;
            }
        }
    }
    if (!copied) {
        long long got = 0;
        while (got < nb) {
            const ssize_t k = pread(fd, static_cast<char*>(dst) + got, (size_t)(nb - got), (off_t)(off + got));
            if (k < 0 && errno == EINTR) continue;
            if (k <= 0) {
                rc = k < 0 ? errno : EIO;
                break;
            }
            got += k;
        }
    }
    close(fd);
    return rc;
}

// The pool lives for the process (never destroyed: its threads are detached and may be parked in
// wait() when the interpreter exits).
struct Pool {
    std::mutex m;
    std::condition_variable cv;
    std::deque<ReadBatch*> active;
    int threads = 0;

    void worker() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return !active.empty(); });
            ReadBatch* b = active.front();
            const int i = b->next++;
            if (b->next >= b->n) active.pop_front();  // the last item is taken: nobody looks at b's queue entry again
            if (i >= b->n) continue;
            lk.unlock();
            b->status[i] = read_one(b->paths[i], b->offsets[i], b->nbytes[i], b->dst[i], b->mmap_min);
            if (b->left.fetch_sub(1) == 1) {
                std::lock_guard<std::mutex> g(b->m);
                b->done = true;
                b->cv.notify_one();
            }
            lk.lock();
        }
    }

    // Returns the number of live workers (0: none could be started -- the caller reads inline).
    int grow(int want) {
        std::lock_guard<std::mutex> g(m);
        while (threads < want) {
            try {
                std::thread(&Pool::worker, this).detach();
            } catch (const std::system_error&) {  // thread limit of the process / cgroup reached
                break;                            // (nothing may cross the extern "C" boundary)
            }
            ++threads;
        }
        return threads;
    }

    void run(ReadBatch& b) {
        {
            std::lock_guard<std::mutex> g(m);
            active.push_back(&b);
        }
        cv.notify_all();
        std::unique_lock<std::mutex> lk(b.m);
        b.cv.wait(lk, [&] { return b.done; });
    }
};

// One pool per PROCESS: a fork()ed child inherits the counters of its parent's pool but none of
// its threads, so it builds its own (the parent's object is left alone: its mutex may have been
// held at the moment of the fork).
Pool& pool() {
    static std::mutex guard;
    static Pool* p = nullptr;
    static pid_t owner = 0;
    std::lock_guard<std::mutex> g(guard);
    const pid_t me = getpid();
    if (!p || owner != me) {
        p = new Pool;
        owner = me;
    }
    return *p;
}

}  // namespace

extern "C" int setk_host_read_payloads(int n, const char* const* paths, const long long* offsets,
                                       const long long* nbytes, void* const* dst, int n_threads,
                                       long long mmap_min_bytes, int* status) {
    if (n < 0 || (n > 0 && (!paths || !offsets || !nbytes || !dst || !status))) return SETK_ERR_INVALID;
    if (n == 0) return SETK_OK;
    for (int i = 0; i < n; ++i) {
        if (!paths[i] || offsets[i] < 0 || nbytes[i] < 0 || (nbytes[i] > 0 && !dst[i])) return SETK_ERR_INVALID;
        status[i] = 0;
    }
    n_threads = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
    ReadBatch b;
    b.n = n;
    b.paths = paths;
    b.offsets = offsets;
    b.nbytes = nbytes;
    b.dst = dst;
    b.mmap_min = mmap_min_bytes;
    b.status = status;
    b.left.store(n);
    Pool& p = pool();
    if (p.grow(n_threads) < 1) {
        for (int i = 0; i < n; ++i) status[i] = read_one(paths[i], offsets[i], nbytes[i], dst[i], mmap_min_bytes);
        return SETK_OK;
    }
    p.run(b);
    return SETK_OK;
}
