// Latency of "launch an empty kernel, wait for the stream" and of a 4-byte D2H read-back, under the runtime's wait modes:
//   hipcc --offload-arch=gfx950 -O2 tools/microbench_sync/sync_latency.cpp -o /tmp/sync_latency && /tmp/sync_latency [spin]
// (env ROC_ACTIVE_WAIT_TIMEOUT=<us> is read by the runtime itself)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 1024) *p = 1; }
__global__ void spin_kernel(long long ticks) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "spin")) printf("hipSetDeviceFlags(spin): %d\n", (int)hipSetDeviceFlags(hipDeviceScheduleSpin));
    if (argc > 1 && !strcmp(argv[1], "yield")) printf("hipSetDeviceFlags(yield): %d\n", (int)hipSetDeviceFlags(hipDeviceScheduleYield));
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int *d, *h; hipMalloc(&d, 64); hipHostMalloc(&h, 64);
    for (int i = 0; i < 50; ++i) { hipLaunchKernelGGL(empty_kernel, 1, 64, 0, s, d); hipStreamSynchronize(s); }
    auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::vector<double> a, b, c;
    for (int i = 0; i < 400; ++i) {
        double t = now(); hipLaunchKernelGGL(empty_kernel, 1, 64, 0, s, d); hipStreamSynchronize(s); a.push_back(now() - t);
        t = now(); hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); b.push_back(now() - t);
        t = now(); hipLaunchKernelGGL(spin_kernel, 1, 64, 0, s, 10000LL /* 100 us at 100 MHz */); hipStreamSynchronize(s); c.push_back(now() - t);
    }
    printf("empty kernel + sync: median %.1f us | 4-byte D2H + sync: %.1f us | 100-us kernel + sync: %.1f us\n", med(a), med(b), med(c));
    return 0;
}
