// Measures the issue rate of v_fma_f32 vs v_pk_fma_f32 (and v_min3 / v_cmp) on gfx950 so
// the ray sweep's instruction mix can be priced.  Build: hipcc --offload-arch=gfx950 -O3
// tools/microbench_valu.hip -o tools/microbench_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 y0 = {x0, x1}, y1 = {x2, x3}, y2 = {x4, x5}, y3 = {x6, x7}, y4 = y0 + 1.f, y5 = y1 + 1.f, y6 = y2 + 1.f, y7 = y3 + 1.f;
    f2 aa = {a, a}, bb = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x1) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x2) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x3) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x4) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x5) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x6) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x7) : "v"(a), "v"(b));
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y0) : "v"(aa), "v"(bb));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y1) : "v"(aa), "v"(bb));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y2) : "v"(aa), "v"(bb));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y3) : "v"(aa), "v"(bb));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y4) : "v"(aa), "v"(bb));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y5) : "v"(aa), "v"(bb));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y6) : "v"(aa), "v"(bb));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y7) : "v"(aa), "v"(bb));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(x1) : "v"(a), "v"(b));
                asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(x2) : "v"(a), "v"(b));
                asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(x3) : "v"(a), "v"(b));
                asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(x4) : "v"(a), "v"(b));
                asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(x5) : "v"(a), "v"(b));
                asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(x6) : "v"(a), "v"(b));
                asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(x7) : "v"(a), "v"(b));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // fma with one SGPR operand (as the sweep uses)
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x0) : "s"(a), "v"(x1));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x1) : "s"(a), "v"(x2));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x2) : "s"(b), "v"(x3));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x3) : "s"(b), "v"(x4));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x4) : "s"(a), "v"(x5));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x5) : "s"(a), "v"(x6));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x6) : "s"(b), "v"(x7));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x7) : "s"(b), "v"(x0));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0.x + y1.y + y2.x + y3.y + y4.x + y5.y + y6.x + y7.y;
}

template <int MODE>
double run(int blocks, int iters, float *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 16, 0.999f, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, cus, p.clockRate);
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD
        int blocks = cus * wps;             // 256 threads = 4 waves = 1 per SIMD
        float *d; hipMalloc(&d, sizeof(float) * 256 * blocks);
        double m[4] = {run<0>(blocks, iters, d), run<1>(blocks, iters, d), run<2>(blocks, iters, d), run<3>(blocks, iters, d)};
        const char *nm[4] = {"v_fma_f32", "v_pk_fma_f32", "v_min3_f32", "v_fma_f32(sgpr)"};
        for (int i = 0; i < 4; ++i) {
            double instr = (double)iters * 32 * wps;  // wave-instructions per SIMD
            double ns_per = m[i] * 1e6 / instr;
            printf("waves/SIMD %d %-16s %8.3f ms  %.3f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)\n", wps, nm[i], m[i], ns_per, ns_per * 2.4);
        }
        hipFree(d);
    }
    return 0;
}
