// hop_latency.hip -- what ONE dependent global-memory access costs a block on this part, for the access kinds a growth step chains together:
//   * pointer chase through a buffer a PREVIOUS kernel wrote (the tree state / slot table every growth kernel reads first), scalar and vector loads
//   * pointer chase over a large (1 GiB) randomly permuted buffer (row gathers)
//   * agent-scope (sc1) atomic load of a word another kernel wrote, and an agent-scope atomic add (arrival counters), round trip
//   * store + s_waitcnt vmcnt(0) of a write-through (agent-scope) store
// build: hipcc -O3 --offload-arch=gfx950 -o bin/hop_latency hop_latency.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_fill(unsigned *p, const unsigned *src, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = src[i]; }
__global__ void k_chase_vec(const unsigned *p, int hops, long long *out)
{
    unsigned i = threadIdx.x;          // (one lane)
    const long long t0 = wall_clock64();
    for (int h = 0; h < hops; h++) i = p[i];
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}
__global__ void k_chase_scalar(const unsigned *p, int hops, long long *out)
{
    unsigned i = 0;
    const long long t0 = wall_clock64();
    for (int h = 0; h < hops; h++) i = __builtin_amdgcn_readfirstlane(p[__builtin_amdgcn_readfirstlane(i)]);     // uniform address: s_load
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}
__global__ void k_chase_agent(unsigned *p, int hops, long long *out)
{
    unsigned i = 0;
    const long long t0 = wall_clock64();
    for (int h = 0; h < hops; h++) i = __hip_atomic_load(&p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}
__global__ void k_atomic_rt(unsigned *p, int hops, long long *out)
{
    unsigned v = 0;
    const long long t0 = wall_clock64();
    for (int h = 0; h < hops; h++) v = __hip_atomic_fetch_add(&p[v & 1023], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
}
__global__ void k_store_wait(unsigned *p, int hops, long long *out)
{
    const long long t0 = wall_clock64();
    for (int h = 0; h < hops; h++) { __hip_atomic_store(&p[(h * 64) & 65535], (unsigned)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = 0; }
}
int main()
{
    const size_t small_n = 1 << 14, big_n = (size_t)1 << 28;       // 64 KiB, 1 GiB
    std::vector<unsigned> perm(big_n);
    std::mt19937_64 rng(1);
    auto make_cycle = [&](size_t n, size_t stride_min) {   // a random single cycle over n slots
        std::vector<unsigned> order(n);
        for (size_t i = 0; i < n; i++) order[i] = (unsigned)i;
        std::shuffle(order.begin() + 1, order.end(), rng);
        for (size_t i = 0; i + 1 < n; i++) perm[order[i]] = order[i + 1];
        perm[order[n - 1]] = order[0];
    };
    unsigned *d_src, *d_small, *d_big; long long *d_out, h_out[2];
    CK(hipMalloc(&d_src, big_n * 4)); CK(hipMalloc(&d_small, small_n * 4)); CK(hipMalloc(&d_big, big_n * 4)); CK(hipMalloc(&d_out, 16));
    const int hops = 256;
    auto report = [&](const char *what) { hipMemcpy(h_out, d_out, 16, hipMemcpyDeviceToHost); printf("%-72s %7.0f ns per hop\n", what, (double)h_out[0] * 10.0 / hops); return 0; };
    // small buffer, rewritten by a wide kernel before every measurement (so it sits in other XCDs' L2s / memory, not in the reader's cache)
    make_cycle(small_n, 0);
    CK(hipMemcpy(d_src, perm.data(), small_n * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, d_small, (const unsigned *)d_src, small_n);
        hipLaunchKernelGGL(k_chase_vec, dim3(1), dim3(1), 0, 0, (const unsigned *)d_small, hops, d_out); CK(hipDeviceSynchronize());
        if (rep) report("vector load chase, 64 KiB buffer just written by another kernel");
        hipLaunchKernelGGL(k_chase_vec, dim3(1), dim3(1), 0, 0, (const unsigned *)d_small, hops, d_out); CK(hipDeviceSynchronize());
        if (rep) report("vector load chase, same buffer again (warm: the reader's own L2 / TCP)");
        hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, d_small, (const unsigned *)d_src, small_n);
        hipLaunchKernelGGL(k_chase_scalar, dim3(1), dim3(64), 0, 0, (const unsigned *)d_small, hops, d_out); CK(hipDeviceSynchronize());
        if (rep) report("scalar (uniform) load chase, 64 KiB buffer just written by another kernel");
        hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, d_small, (const unsigned *)d_src, small_n);
        hipLaunchKernelGGL(k_chase_agent, dim3(1), dim3(1), 0, 0, d_small, hops, d_out); CK(hipDeviceSynchronize());
        if (rep) report("agent-scope atomic load chase, 64 KiB buffer just written by another kernel");
        hipLaunchKernelGGL(k_chase_agent, dim3(1), dim3(1), 0, 0, d_small, hops, d_out); CK(hipDeviceSynchronize());
        if (rep) report("agent-scope atomic load chase, same buffer again");
        CK(hipMemset(d_small, 0, small_n * 4));
        hipLaunchKernelGGL(k_atomic_rt, dim3(1), dim3(1), 0, 0, d_small, hops, d_out); CK(hipDeviceSynchronize());
        if (rep) report("agent-scope atomic fetch-add, returned value needed (round trip)");
        hipLaunchKernelGGL(k_store_wait, dim3(1), dim3(1), 0, 0, d_small, hops, d_out); CK(hipDeviceSynchronize());
        if (rep) report("agent-scope store + s_waitcnt vmcnt(0)");
    }
    make_cycle(big_n, 0);
    CK(hipMemcpy(d_src, perm.data(), big_n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, d_big, (const unsigned *)d_src, big_n);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_chase_vec, dim3(1), dim3(1), 0, 0, (const unsigned *)d_big, hops, d_out); CK(hipDeviceSynchronize());
        if (rep) report("vector load chase, 1 GiB random cycle (HBM, TLB misses)");
        hipLaunchKernelGGL(k_chase_agent, dim3(1), dim3(1), 0, 0, d_big, hops, d_out); CK(hipDeviceSynchronize());
        if (rep) report("agent-scope atomic load chase, 1 GiB random cycle");
    }
    return 0;
}
