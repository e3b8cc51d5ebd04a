// Dependent f64 addition chains on gfx950: cycles per step of  s = s + x  (what bounds RL_FLAG_JAVA_ORDER's per-bin chains).
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/micro/f64_chain.hip -o /tmp/f64_chain && /tmp/f64_chain
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_chain(const double *x, int n, int chains, double *out, long long *cyc)
{
    __shared__ double lx[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lx[i] = x[i];
    __syncthreads();
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    const long long t0 = clock64();
    for (int r = 0; r < n; r += 2048)
        for (int j = 0; j < 2048; j += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = lx[j + u];
            if (chains == 1) {
#pragma unroll
                for (int u = 0; u < 8; u++) s0 += v[u];
            } else if (chains == 8) {               // the same chain through v_fma_f64: fma(s, 1.0, x) == s + x exactly
#pragma unroll
                for (int u = 0; u < 8; u++) asm volatile("v_fma_f64 %0, %0, 1.0, %1" : "+v"(s0) : "v"(v[u]));
            } else if (chains == 16) {              // f32 chain for comparison
                float f = (float)s0;
#pragma unroll
                for (int u = 0; u < 8; u++) f += (float)v[u];
                s0 = f;
            } else if (chains == 2) {
#pragma unroll
                for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
            } else {
#pragma unroll
                for (int u = 0; u < 8; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
            }
        }
    const long long t1 = clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = s0 + s1 + s2 + s3; cyc[blockIdx.x] = t1 - t0; }
}
int main()
{
    double *x, *out; long long *cyc;
    hipMalloc(&x, 2048 * 8); hipMalloc(&out, 8 * 64); hipMalloc(&cyc, 8 * 64);
    double h[2048]; for (int i = 0; i < 2048; i++) h[i] = 1e-3 * (i % 17) - 7e-3;
    hipMemcpy(x, h, sizeof(h), hipMemcpyHostToDevice);
    const int n = 2048 * 512;
    for (int waves = 1; waves <= 16; waves *= 4)
        for (int chains = 1; chains <= 16; chains *= 2) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k_chain, dim3(1), dim3(64 * waves), 0, 0, x, n, chains, out, cyc);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_chain, dim3(1), dim3(64 * waves), 0, 0, x, n, chains, out, cyc);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("waves/block %2d, variant %2d: %.2f clock64 ticks, %.2f ns per element\n", waves, chains, (double)c / n, ms * 1e6 / n);
        }
    return 0;
}
