// Microbenchmark: cost of one step of the float running sum  s = (float)((double)s + x)  in the forms the chain kernels
// could use (tools/README.md).  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off chain_step.hip -o chain_step && ./chain_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ double bits2d(uint64_t u) { return __longlong_as_double((long long)u); }
__device__ __forceinline__ uint64_t d2bits(double d) { return (uint64_t)__double_as_longlong(d); }
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

template <bool VELT> struct State;
template <> struct State<false> {
    float s;
    __device__ __forceinline__ void init(float v) { s = v; }
    __device__ __forceinline__ void step(double x) { s = (float)((double)s + x); }
    __device__ __forceinline__ float get() const { return s; }
};
template <> struct State<true> {
    double s;
    __device__ __forceinline__ void init(float v) { s = v; }
    __device__ __forceinline__ void step(double x) { const double y = s + x, p = y * 536870913.0, q = y - p; s = __builtin_copysign(q + p, y); }
    __device__ __forceinline__ float get() const { return (float)s; }
};

template <int J> __device__ __forceinline__ int row_share(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xf, 0xf, false); }

// MODE 0: ds_bpermute broadcast inside groups of GL lanes   1: DPP row_share (GL = 16)   2: wave-uniform scalar loads   3: no loads
template <int MODE, bool VELT, int GL>
__global__ __launch_bounds__(256) void k_steps(const double *x, int n, float *out)
{
    State<VELT> st;
    st.init((float)(threadIdx.x & 31) * 1e-3f);
    const int gl = lane_id() & (GL - 1), gbase = lane_id() & ~(GL - 1);
    if (MODE == 0 || MODE == 1) {
        double nxt = x[gl];
        for (int i0 = 0; i0 < n; i0 += GL) {
            const double mine = nxt;
            nxt = (i0 + GL + gl < n) ? x[i0 + GL + gl] : 0.0;
            const int lo = (int)(uint32_t)d2bits(mine), hi = (int)(uint32_t)(d2bits(mine) >> 32);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < GL; j++) {
                    const uint32_t l2 = (uint32_t)__shfl(lo, gbase + j), h2 = (uint32_t)__shfl(hi, gbase + j);
                    st.step(bits2d(((uint64_t)h2 << 32) | l2));
                }
            } else {
#define RS(J) { const uint32_t l2 = (uint32_t)row_share<J>(lo), h2 = (uint32_t)row_share<J>(hi); st.step(bits2d(((uint64_t)h2 << 32) | l2)); }
                RS(0) RS(1) RS(2) RS(3) RS(4) RS(5) RS(6) RS(7) RS(8) RS(9) RS(10) RS(11) RS(12) RS(13) RS(14) RS(15)
            }
        }
    } else if (MODE == 2) {
        const double *xu = x + __builtin_amdgcn_readfirstlane((int)(blockIdx.x & 7) * 8);
        for (int i0 = 0; i0 < n; i0 += 16) {
#pragma unroll
            for (int j = 0; j < 16; j++) st.step(xu[i0 + j]);
        }
    } else {
        const double c = x[0];
        for (int i0 = 0; i0 < n; i0++) st.step(c);
    }
    out[blockIdx.x * 256 + threadIdx.x] = st.get();
}

template <int MODE, bool VELT, int GL>
static int run(const char *name, const double *dx, int n, float *dout, int blocks, float *ref)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_steps<MODE, VELT, GL>), dim3(blocks), dim3(256), 0, 0, dx, n, dout);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL((k_steps<MODE, VELT, GL>), dim3(blocks), dim3(256), 0, 0, dx, n, dout);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    float h[64]; CK(hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost));
    bool same = true;
    if (MODE < 2) { if (ref[0] == -12345.f) for (int i = 0; i < 64; i++) ref[i] = h[i]; else for (int i = 0; i < 64; i++) same = same && (h[i] == ref[i]); }
    printf("%-44s %8.1f us per launch  %6.1f ns per step  (%.1f SIMD cycles per wave-step at %d waves/SIMD, 2.4 GHz)%s\n", name, ms * 100.0, ms * 1e5 / n,
           ms * 1e5 / n * 2.4 / (blocks * 4 / 1024.0), blocks * 4 / 1024, same ? "" : "  RESULT MISMATCH");
    return 0;
}

int main()
{
    const int n = 1024;
    std::vector<double> hx(n + 64);
    uint64_t sd = 88172645463325252ull;
    for (auto &v : hx) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; v = ((double)(sd >> 11) / 9007199254740992.0 - 0.5) * 1e-2; }
    double *dx; float *dout;
    CK(hipMalloc((void **)&dx, hx.size() * 8)); CK(hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
    for (int blocks : {256 * 2, 256 * 4, 256 * 8}) {
        CK(hipMalloc((void **)&dout, (size_t)blocks * 256 * 4));
        float ref[64]; ref[0] = -12345.f;
        printf("---- %d blocks of 256 threads (%d waves per SIMD), %d steps\n", blocks, blocks * 4 / 1024, n);
        if (run<0, false, 16>("bpermute x2 (GL 16) + conversions", dx, n, dout, blocks, ref)) return 1;
        if (run<0, true, 16>("bpermute x2 (GL 16) + Veltkamp", dx, n, dout, blocks, ref)) return 1;
        if (run<0, false, 32>("bpermute x2 (GL 32) + conversions", dx, n, dout, blocks, ref)) return 1;
        if (run<1, false, 16>("DPP row_share x2 + conversions", dx, n, dout, blocks, ref)) return 1;
        if (run<1, true, 16>("DPP row_share x2 + Veltkamp", dx, n, dout, blocks, ref)) return 1;
        if (run<2, false, 16>("wave-uniform scalar loads + conversions", dx, n, dout, blocks, ref)) return 1;
        if (run<2, true, 16>("wave-uniform scalar loads + Veltkamp", dx, n, dout, blocks, ref)) return 1;
        if (run<3, false, 16>("register operand + conversions", dx, n, dout, blocks, ref)) return 1;
        if (run<3, true, 16>("register operand + Veltkamp", dx, n, dout, blocks, ref)) return 1;
        CK(hipFree(dout));
    }
    return 0;
}
