// Dependent-issue latency of v_add_f64 / v_fma_f64 / v_add_f32 on gfx950: registers only, one wavefront, K independent chains.
// hipcc -O3 --offload-arch=gfx950 tools/micro/f64_latency.hip -o /tmp/f64_latency && /tmp/f64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K, int OP>
__global__ void k_lat(double x0, int iters, double *out, long long *cyc)
{
    double s[4] = {x0, x0 + 1, x0 + 2, x0 + 3};
    double x = x0 * 0.5;
    float f[4] = {1.f, 2.f, 3.f, 4.f}; float xf = (float)x;
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 64 / K; u++)
#pragma unroll
            for (int c = 0; c < K; c++) {
                if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(s[c]) : "v"(x));
                else if (OP == 1) asm volatile("v_fma_f64 %0, %0, 1.0, %1" : "+v"(s[c]) : "v"(x));
                else asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[c]) : "v"(xf));
            }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = s[0] + s[1] + s[2] + s[3] + f[0] + f[1] + f[2] + f[3]; cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
template <int K, int OP> void run(const char *name, double *out, long long *cyc)
{
    const int iters = 20000;
    hipLaunchKernelGGL((k_lat<K, OP>), dim3(1), dim3(64), 0, 0, 1.25, iters, out, cyc);
    hipDeviceSynchronize();
    long long cw[2]; hipMemcpy(cw, cyc, 16, hipMemcpyDeviceToHost);
    const long long c = cw[0];
    // wall_clock64() ticks at 100 MHz: ns per instruction, and the frequency clock64() counts at
    printf("%-12s %d independent chain(s): %.2f cycles per instruction (=> %.2f per dependent step); %.2f ns per instruction, clock64 at %.0f MHz\n", name, K,
           (double)c / (iters * 64.0), (double)c / (iters * 64.0) * K, (double)cw[1] * 10.0 / (iters * 64.0), (double)c / ((double)cw[1] * 10.0) * 1000.0);
}
int main()
{
    double *out; long long *cyc; hipMalloc(&out, 64); hipMalloc(&cyc, 64);
    run<1, 0>("v_add_f64", out, cyc); run<2, 0>("v_add_f64", out, cyc); run<4, 0>("v_add_f64", out, cyc);
    run<1, 1>("v_fma_f64", out, cyc); run<4, 1>("v_fma_f64", out, cyc);
    run<1, 2>("v_add_f32", out, cyc); run<4, 2>("v_add_f32", out, cyc);
    return 0;
}
