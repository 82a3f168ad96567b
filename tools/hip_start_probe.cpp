// development aid: what does starting HIP cost on this machine, without any of this repo's code?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(int* p) { if (p) *p = 1; }
int main() {
    double t0 = now(); int n = 0; hipGetDeviceCount(&n);
    double t1 = now(); hipSetDevice(0); hipFree(0);
    double t2 = now(); hipStream_t s; hipStreamCreate(&s);
    double t3 = now(); int* d; hipMalloc(&d, 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d); hipStreamSynchronize(s);
    double t4 = now();
    printf("hipGetDeviceCount %.1f ms, hipSetDevice+hipFree(0) %.1f ms, stream %.1f ms, first kernel %.1f ms\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    return 0;
}
