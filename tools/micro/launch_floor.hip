// launch_floor.hip -- cost of dependent kernel boundaries on one stream (eager launches vs hipGraph replay).
// build: hipcc -O3 --offload-arch=gfx950 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { int v[80]; };
__global__ void k_noop(const int *flag, int *out, Big b) { if (*flag) out[blockIdx.x] = b.v[threadIdx.x & 63]; }
__global__ void k_touch(const int *flag, int *out, Big b) { out[blockIdx.x * 256 + threadIdx.x] = b.v[3] + *flag; }
__global__ void k_spin(int *out, long long cycles) { const long long t0 = clock64(); while (clock64() - t0 < cycles) ; if (cycles == 1) out[0] = 1; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    int *flag, *out; CK(hipMalloc(&flag, 4)); CK(hipMalloc(&out, 4 << 20)); CK(hipMemset(flag, 0, 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Big b = {};
    const int n = 2000;
    for (int grid : {1, 256, 2048, 8192}) {
        for (int variant = 0; variant < 2; variant++) {
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < n; i++) {
                    if (variant == 0) hipLaunchKernelGGL(k_noop, dim3(grid), dim3(256), 0, s, (const int *)flag, out, b);
                    else hipLaunchKernelGGL(k_touch, dim3(grid > 2048 ? 2048 : grid), dim3(256), 0, s, (const int *)flag, out, b);
                }
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("eager %-6s grid %5d: %.2f us per launch\n", variant ? "touch" : "noop", grid, 1e3 * ms / n);
            }
        }
    }
    // device-bound eager floor: the launches queue up behind a long-running kernel, so the host is far ahead
    for (int variant = 0; variant < 2; variant++) {
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, out, 40000000LL);     // ~20 ms
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < n; i++) {
                if (variant == 0 || (i & 1)) hipLaunchKernelGGL(k_noop, dim3(256), dim3(256), 0, s, (const int *)flag, out, b);
                else hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, s, (const int *)flag, out, b);
            }
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("eager queued-behind-spin %s: %.2f us per launch\n", variant ? "alternating 2 kernels" : "one kernel", 1e3 * ms / n);
        }
    }
    // graph replay of 200 no-op launches
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_noop, dim3(256), dim3(256), 0, s, (const int *)flag, out, b);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 10; i++) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("graph noop grid 256: %.2f us per launch\n", 1e3 * ms / 2000);
    }
    return 0;
}
