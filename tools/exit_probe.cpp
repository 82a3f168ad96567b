// development aid: what does a process that used the GPU pay at exit?  The child sets up one thing (device memory, pinned memory, streams, a touched
// file mapping), stamps the clock and calls _exit; the parent measures stamp -> reaped.  usage: exit_probe <file for the mapping> (hipcc exit_probe.cpp)
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
struct Case { const char* name; double vram_gb; int pinned_mb; int pinned_pieces; int streams; double map_gb; int use_streams; };
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : nullptr;
    const Case cases[] = {
        {"HIP start-up only", 0, 0, 0, 0, 0, 0},
        {"10 GB of device memory, written", 10, 0, 0, 0, 0, 0},
        {"256 MB pinned in 32 pieces", 0, 256, 32, 0, 0, 0},
        {"32 MB pinned in 16 pieces", 0, 32, 16, 0, 0, 0},
        {"16 streams, created only", 0, 0, 0, 16, 0, 0},
        {"16 streams, a copy on each", 0, 64, 1, 16, 0, 1},
        {"4 streams, a copy on each", 0, 64, 1, 4, 0, 1},
        {"6 GB of a file mapping touched", 0, 0, 0, 0, 6, 0},
        {"all of it", 10, 256, 32, 16, 6, 1},
        {"HIP start-up only (again)", 0, 0, 0, 0, 0, 0},
    };
    for (const Case& c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            int pfd[2]; if (pipe(pfd)) return 1;
            const double t_fork = now();
            pid_t pid = fork();
            if (pid == 0) {
                close(pfd[0]);
                hipSetDevice(0);
                hipFree(nullptr);
                void* d = nullptr;
                if (c.vram_gb > 0) { hipMalloc(&d, (size_t)(c.vram_gb * 1e9)); hipMemset(d, 1, (size_t)(c.vram_gb * 1e9)); hipDeviceSynchronize(); }
                std::vector<void*> pins;
                for (int i = 0; i < c.pinned_pieces; ++i) { void* p = nullptr; hipHostMalloc(&p, (size_t)c.pinned_mb * (1u << 20) / c.pinned_pieces, hipHostMallocDefault); memset(p, 1, 4096); pins.push_back(p); }
                std::vector<hipStream_t> st((size_t)c.streams);
                void* dd = nullptr; if (c.use_streams) hipMalloc(&dd, 64u << 20);
                for (auto& s : st) { hipStreamCreateWithFlags(&s, hipStreamNonBlocking); if (c.use_streams && !pins.empty()) hipMemcpyAsync(dd, pins[0], 1u << 20, hipMemcpyHostToDevice, s); }
                hipDeviceSynchronize();
                unsigned long acc = 0;
                if (c.map_gb > 0 && path) {
                    int fd = open(path, O_RDONLY); struct stat sb; fstat(fd, &sb);
                    const size_t n = (size_t)sb.st_size;
                    const char* m = (const char*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
                    for (size_t i = 0; i < n && i < (size_t)(c.map_gb * 1e9); i += 4096) acc += m[i];
                }
                const double t = now() + (acc == 12345 ? 1 : 0);
                if (write(pfd[1], &t, sizeof t) != sizeof t) _exit(3);
                _exit(0);
            }
            close(pfd[1]);
            double t_stamp = 0;
            if (read(pfd[0], &t_stamp, sizeof t_stamp) != sizeof t_stamp) t_stamp = 0;
            int stt = 0; waitpid(pid, &stt, 0);
            const double t_reaped = now();
            close(pfd[0]);
            printf("%-36s run %d: set-up %.0f ms, exit (stamp -> reaped) %.1f ms\n", c.name, rep, t_stamp - t_fork, t_reaped - t_stamp);
            fflush(stdout);
        }
    }
    return 0;
}
