// shm_io — what does it cost to get N GB from HBM into a file under /dev/shm on this box?
// (the BAM hand-off of stage 1: TELR_alignment.py:103-114).  Measures
//   1. D2H into pinned staging (hipMemcpyAsync), one stream
//   2. pwrite of the staging buffer into a fresh tmpfs file with T threads
//   3. memcpy into a fresh MAP_SHARED mapping of a tmpfs file with T threads (page faults included)
//   4. hipHostRegister of such a mapping + D2H straight into it (if the driver accepts file-backed pages)
// build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/shm_io tools/ubench/shm_io.hip -lpthread
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char **argv)
{
    const size_t GB = (size_t)1 << 30;
    const size_t total = (argc > 1 ? (size_t)atol(argv[1]) : 4) * GB;
    const size_t chunk = 256u << 20;
    const char *path = "/dev/shm/telr_shm_io.bin";
    void *d = nullptr; CK(hipMalloc(&d, total)); CK(hipMemset(d, 0x5a, total));
    void *h = nullptr; CK(hipHostMalloc(&h, chunk * 2, hipHostMallocDefault));
    hipStream_t s; CK(hipStreamCreate(&s));
    // 1. D2H pinned
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (size_t o = 0; o < total; o += chunk) CK(hipMemcpyAsync((char*)h + (o / chunk % 2) * chunk, (char*)d + o, chunk, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double dt = now() - t0;
        printf("D2H pinned ring: %.2f GB in %.3f s = %.1f GB/s\n", total / 1e9, dt, total / 1e9 / dt);
    }
    // 2. pwrite with T threads (fresh file each time)
    for (int T : {1, 4, 8, 16, 32}) {
        unlink(path);
        int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600);
        if (fd < 0) { perror("open"); return 1; }
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
            const size_t piece = 8u << 20;
            for (size_t o = (size_t)t * piece; o < total; o += (size_t)T * piece) {
                size_t n = std::min(piece, total - o);
                if (pwrite(fd, (char*)h + (o % chunk), n, (off_t)o) != (ssize_t)n) { perror("pwrite"); break; }
            }
        });
        for (auto &x : th) x.join();
        double dt = now() - t0;
        printf("pwrite to tmpfs, %2d threads: %.2f GB in %.3f s = %.1f GB/s\n", T, total / 1e9, dt, total / 1e9 / dt);
        close(fd);
    }
    // 2b. pwrite over an EXISTING file (pages already allocated)
    for (int T : {8, 16}) {
        int fd = open(path, O_RDWR, 0600);
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
            const size_t piece = 8u << 20;
            for (size_t o = (size_t)t * piece; o < total; o += (size_t)T * piece) { size_t n = std::min(piece, total - o); if (pwrite(fd, (char*)h + (o % chunk), n, (off_t)o) != (ssize_t)n) break; }
        });
        for (auto &x : th) x.join();
        double dt = now() - t0;
        printf("pwrite over existing tmpfs pages, %2d threads: %.1f GB/s\n", T, total / 1e9 / dt);
        close(fd);
    }
    // 3. memcpy into a fresh mapping
    for (int T : {8, 16}) {
        unlink(path);
        int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600);
        if (ftruncate(fd, (off_t)total) != 0) { perror("ftruncate"); return 1; }
        char *m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { perror("mmap"); return 1; }
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
            const size_t piece = 8u << 20;
            for (size_t o = (size_t)t * piece; o < total; o += (size_t)T * piece) memcpy(m + o, (char*)h + (o % chunk), std::min(piece, total - o));
        });
        for (auto &x : th) x.join();
        double dt = now() - t0;
        printf("memcpy into fresh MAP_SHARED tmpfs mapping, %2d threads: %.1f GB/s\n", T, total / 1e9 / dt);
        munmap(m, total); close(fd);
    }
    // 4. register a mapping and DMA into it
    {
        unlink(path);
        int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600);
        if (ftruncate(fd, (off_t)total) != 0) { perror("ftruncate"); return 1; }
        char *m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        double t0 = now();
        hipError_t e = hipHostRegister(m, total, hipHostRegisterDefault);
        double t_reg = now() - t0;
        printf("hipHostRegister(tmpfs mapping, %.1f GB): %s, %.3f s\n", total / 1e9, hipGetErrorString(e), t_reg);
        if (e == hipSuccess) {
            t0 = now();
            CK(hipMemcpyAsync(m, d, total, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            double dt = now() - t0;
            printf("D2H straight into the registered mapping: %.1f GB/s (register + copy: %.1f GB/s)\n", total / 1e9 / dt, total / 1e9 / (dt + t_reg));
            bool ok = m[0] == 0x5a && m[total - 1] == 0x5a && m[total / 2] == 0x5a;
            printf("content %s\n", ok ? "ok" : "WRONG");
            t0 = now(); hipHostUnregister(m); printf("unregister %.3f s\n", now() - t0);
        } else (void)hipGetLastError();
        munmap(m, total); close(fd);
    }
    // 6. ways of getting the pages of a tmpfs file allocated (the writer's real cost)
    {
        unlink(path);
        int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600);
        double t0 = now();
        int e = posix_fallocate(fd, 0, (off_t)total);
        double dt = now() - t0;
        printf("posix_fallocate %.1f GB: rc %d, %.3f s = %.1f GB/s\n", total / 1e9, e, dt, total / 1e9 / dt);
        for (int T : {1, 2, 4}) {
            t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                const size_t piece = 32u << 20;
                for (size_t o = (size_t)t * piece; o < total; o += (size_t)T * piece) { size_t n = std::min(piece, total - o); if (pwrite(fd, (char*)h + (o % chunk), n, (off_t)o) != (ssize_t)n) break; }
            });
            for (auto &x : th) x.join();
            dt = now() - t0;
            printf("pwrite over fallocated pages, %d thread(s), 32-MB pieces: %.1f GB/s\n", T, total / 1e9 / dt);
        }
        close(fd); unlink(path);
        // one thread touches pages ahead through a mapping (no inode lock) while one thread pwrites behind it
        fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600);
        if (ftruncate(fd, (off_t)total) != 0) return 1;
        char *m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        t0 = now();
        {
            double tp0 = now();
#ifdef MADV_POPULATE_WRITE
            int rc = madvise(m, total, MADV_POPULATE_WRITE);
#else
            int rc = -1;
#endif
            printf("madvise(MADV_POPULATE_WRITE) on a fresh tmpfs mapping: rc %d, %.3f s = %.1f GB/s\n", rc, now() - tp0, total / 1e9 / (now() - tp0));
        }
        munmap(m, total); close(fd); unlink(path);
        // two files' worth? no: ONE file, first half written by thread A while thread B writes the second half
        fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600);
        t0 = now();
        {
            std::vector<std::thread> th;
            for (int t = 0; t < 2; ++t) th.emplace_back([&, t] {
                const size_t piece = 32u << 20, lo = t * (total / 2), hi = t ? total : total / 2;
                for (size_t o = lo; o < hi; o += piece) { size_t n = std::min(piece, hi - o); if (pwrite(fd, (char*)h + (o % chunk), n, (off_t)o) != (ssize_t)n) break; }
            });
            for (auto &x : th) x.join();
        }
        dt = now() - t0;
        printf("pwrite to a fresh file, 2 threads on the two halves: %.1f GB/s\n", total / 1e9 / dt);
        close(fd); unlink(path);
        // write() appends from one thread in 1-MB, 8-MB, 64-MB calls
        for (size_t piece : {(size_t)1 << 20, (size_t)8 << 20, (size_t)64 << 20}) {
            fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600);
            t0 = now();
            for (size_t o = 0; o < total; o += piece) { size_t n = std::min(piece, total - o); if (write(fd, (char*)h + (o % chunk), n) != (ssize_t)n) break; }
            dt = now() - t0;
            printf("write() append, one thread, %zu-MB calls: %.1f GB/s\n", piece >> 20, total / 1e9 / dt);
            close(fd); unlink(path);
        }
    }
    // 7. fallocate, then memcpy through a mapping of the ALLOCATED file with T threads (no inode lock on this path)
    for (int T : {1, 4, 8, 16}) {
        unlink(path);
        int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600);
        double t0 = now();
        if (posix_fallocate(fd, 0, (off_t)total) != 0) return 1;
        double t_fa = now() - t0;
        char *m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
            const size_t piece = 8u << 20;
            for (size_t o = (size_t)t * piece; o < total; o += (size_t)T * piece) memcpy(m + o, (char*)h + (o % chunk), std::min(piece, total - o));
        });
        for (auto &x : th) x.join();
        double dt = now() - t0;
        printf("fallocate (%.3f s) + memcpy into the mapping of the allocated file, %2d threads: %.1f GB/s\n", t_fa, T, total / 1e9 / dt);
        munmap(m, total); close(fd);
    }
    // 8. fallocate, mmap, pre-fault the mapping (MADV_POPULATE_WRITE with P threads on slices; the pages exist, only the page
    //    table is filled), then memcpy with 16 threads: is the 14.6 GB/s above the fault path or the memory?
    if (argc > 2) for (int P : {1, 4, 8}) {
        unlink(path);
        int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600);
        double t0 = now();
        if (posix_fallocate(fd, 0, (off_t)total) != 0) return 1;
        double t_fa = now() - t0;
        char *m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        t0 = now();
        {
            std::vector<std::thread> th; int rc_all = 0;
            for (int t = 0; t < P; ++t) th.emplace_back([&, t] {
                size_t lo = total / P * t, hi = t == P - 1 ? total : total / P * (t + 1);
                if (madvise(m + lo, hi - lo, 23 /* MADV_POPULATE_WRITE */) != 0) rc_all = 1;
            });
            for (auto &x : th) x.join();
            if (rc_all) printf("populate failed\n");
        }
        double t_pop = now() - t0;
        for (int T : {4, 16}) {
            t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                const size_t piece = 8u << 20;
                for (size_t o = (size_t)t * piece; o < total; o += (size_t)T * piece) memcpy(m + o, (char*)h + (o % chunk), std::min(piece, total - o));
            });
            for (auto &x : th) x.join();
            double dt = now() - t0;
            printf("fallocate (%.3f s) + populate with %d thread(s) (%.3f s) + memcpy into the populated mapping, %2d threads: %.1f GB/s\n", t_fa, P, t_pop, T, total / 1e9 / dt);
        }
        munmap(m, total); close(fd);
    }
    // 9. the same without fallocate: populate allocates, zeroes and maps (P threads on slices)
    if (argc > 2) for (int P : {1, 4, 8}) {
        unlink(path);
        int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600);
        if (ftruncate(fd, (off_t)total) != 0) return 1;
        char *m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        double t0 = now();
        {
            std::vector<std::thread> th;
            for (int t = 0; t < P; ++t) th.emplace_back([&, t] {
                size_t lo = total / P * t, hi = t == P - 1 ? total : total / P * (t + 1);
                madvise(m + lo, hi - lo, 23);
            });
            for (auto &x : th) x.join();
        }
        double t_pop = now() - t0;
        printf("ftruncate + populate (allocates) with %d thread(s): %.3f s = %.1f GB/s\n", P, t_pop, total / 1e9 / t_pop);
        munmap(m, total); close(fd);
    }
    unlink(path);
    // 5. host-side deflate / crc32 rates are measured by the library's own writer (bench.py --bam-leg host)
    hipFree(d); hipHostFree(h);
    return 0;
}
