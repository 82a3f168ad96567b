// development aid: how fast can this machine get a page-cached file into a process?  mmap first touch vs pread, 1 / N threads;
// and write() vs parallel pwrite() of a buffer.  usage: io_probe <file>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <algorithm>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* path = argv[1];
    int fd = open(path, O_RDONLY); struct stat st; fstat(fd, &st); const size_t n = st.st_size;
    for (int T : {1, 4, 16}) {
        for (int pop = 0; pop < 2; ++pop) {
            double t0 = now();
            char* m = (char*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE | (pop ? MAP_POPULATE : 0), fd, 0);
            double t1 = now();
            std::vector<std::thread> th; std::vector<unsigned long> acc(T * 8);
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] { size_t lo = n / T * t, hi = t == T - 1 ? n : n / T * (t + 1); unsigned long s = 0;
                for (size_t i = lo; i < hi; i += 4096) s += m[i]; acc[t * 8] = s; });
            for (auto& x : th) x.join();
            double t2 = now();
            th.clear();
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] { size_t lo = n / T * t, hi = t == T - 1 ? n : n / T * (t + 1); unsigned long s = 0;
                for (size_t i = lo; i < hi;) { const char* e = (const char*)memchr(m + i, '\n', hi - i); if (!e) break; s += (e - m); i = (e - m) + 1; } acc[t * 8] = s; });
            for (auto& x : th) x.join();
            double t3 = now();
            munmap(m, n);
            printf("mmap%s T=%2d: map %.1f ms, first touch %.1f ms, memchr pass %.1f ms\n", pop ? "+POPULATE" : "", T, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
        }
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] { size_t lo = n / T * t, hi = t == T - 1 ? n : n / T * (t + 1); std::vector<char> buf(1 << 20);
            for (size_t i = lo; i < hi;) { ssize_t r = pread(fd, buf.data(), std::min(buf.size(), hi - i), i); if (r <= 0) break; i += r; } });
        for (auto& x : th) x.join();
        printf("pread 1 MiB chunks T=%2d: %.1f ms\n", T, (now() - t0) * 1e3);
    }
    // writing
    std::vector<char> buf(size_t(320) << 20, 'x');
    std::string o = std::string(path) + ".probe_out";
    for (int T : {1, 4}) {
        int ofd = open(o.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] { size_t lo = buf.size() / T * t, hi = t == T - 1 ? buf.size() : buf.size() / T * (t + 1);
            for (size_t i = lo; i < hi;) { ssize_t r = pwrite(ofd, buf.data() + i, std::min<size_t>(16 << 20, hi - i), i); if (r <= 0) break; i += r; } });
        for (auto& x : th) x.join();
        double t1 = now(); close(ofd);
        printf("pwrite 320 MiB T=%d: %.1f ms (close %.1f ms)\n", T, (t1 - t0) * 1e3, (now() - t1) * 1e3);
    }
    unlink(o.c_str());
    return 0;
}
