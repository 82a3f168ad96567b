// development aid: is write(2) from hipHostMalloc'd memory slower than from malloc'd memory on this machine?
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double wr(const char* path, const char* buf, size_t chunk, size_t total) {
    int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    double t0 = now();
    for (size_t off = 0; off < total; off += chunk) { size_t n = chunk < total - off ? chunk : total - off; if (write(fd, buf, n) != (ssize_t)n) break; }
    double t1 = now(); close(fd); unlink(path); return t1 - t0;
}
int main() {
    const size_t chunk = 4u << 20, total = 320u << 20;
    char* d; hipMalloc(&d, chunk); hipMemset(d, 7, chunk);
    char* m = (char*)malloc(chunk); memset(m, 7, chunk);
    for (unsigned flags : {(unsigned)hipHostMallocDefault, (unsigned)hipHostMallocNonCoherent, (unsigned)hipHostMallocCoherent}) {
        char* p = nullptr; double t0 = now();
        if (hipHostMalloc((void**)&p, chunk, flags) != hipSuccess) { printf("flags %u: alloc failed\n", flags); continue; }
        double t1 = now(); hipMemcpy(p, d, chunk, hipMemcpyDeviceToHost);
        double t2 = now();
        volatile unsigned long s = 0; for (size_t i = 0; i < chunk; i += 8) s += *(unsigned long*)(p + i);
        double t3 = now();
        printf("hipHostMalloc flags %u: alloc %.2f ms, D2H 4 MiB %.2f ms, CPU read 4 MiB %.2f ms, write 320 MiB from it %.1f ms\n", flags, t1 - t0, t2 - t1, t3 - t2,
               wr("/tmp/pwp.out", p, chunk, total));
        hipHostFree(p);
    }
    { volatile unsigned long s = 0; double t2 = now(); for (size_t i = 0; i < chunk; i += 8) s += *(unsigned long*)(m + i); double t3 = now();
      printf("malloc: CPU read 4 MiB %.2f ms, write 320 MiB from it %.1f ms\n", t3 - t2, wr("/tmp/pwp.out", m, chunk, total)); }
    hipHostRegister(m, chunk, hipHostRegisterDefault);
    double t0 = now(); hipMemcpy(m, d, chunk, hipMemcpyDeviceToHost); double t1 = now();
    printf("malloc + hipHostRegister: D2H 4 MiB %.2f ms, write 320 MiB from it %.1f ms\n", t1 - t0, wr("/tmp/pwp.out", m, chunk, total));
    return 0;
}
