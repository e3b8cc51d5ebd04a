// How fast can ONE wavefront run  s = s + x[j]  over x in LDS, in order?  (RL_FLAG_JAVA_ORDER: k_jhist2's walk.)
// Variants: 0 = every lane reads the same element (LDS broadcast read, 3 batches of 8 in flight)      -- jwalk as shipped
//           1 = the same, only lane 0 active (exec = 1)
//           2 = lane l reads element 64 k + l once, elements reach the chain through v_readlane (SGPR operand)
//           3 = as 2, the chain runs on lane 0 only
//           5 = FOUR chains per wavefront, one per row of 16 lanes: lane l reads element 16 k + (l & 15) of its row's list once, the
//               row's lanes take element u through v_mov_b64_dpp row_newbcast:u (64-bit DPP knows only this pattern; v_add_f64 has no DPP form)
//           6 = as 5, every element owned by ONE lane of the row (16 bins per row): s += (owner == lane) ? x : 0
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/micro/f64_walk.hip -o tools/micro/bin/f64_walk
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int T = 2048;
template <int V>
__global__ void k_walk(const double *x, int reps, double *out, long long *cyc)
{
    __shared__ double lx[T];
    for (int i = threadIdx.x; i < T; i += blockDim.x) lx[i] = x[i];
    __syncthreads();
    double s = 0.0;
    const int lane = threadIdx.x;
    const long long t0 = clock64();
    if (V == 0 || V == 1 || V == 4) {
        if (V != 1 || lane == 0) {
            for (int r = 0; r < reps; r++) {
                double x0[8], x1[8], x2[8], x3[8];
#define LD(X, b) _Pragma("unroll") for (int u = 0; u < 8; u++) X[u] = lx[(b) + u];
#define AD(X) _Pragma("unroll") for (int u = 0; u < 8; u++) s += X[u];
                int j = 0;
                LD(x0, 0) LD(x1, 8) LD(x2, 16)
                while (j + 56 <= T) {
                    LD(x3, j + 24) AD(x0)
                    LD(x0, j + 32) AD(x1)
                    LD(x1, j + 40) AD(x2)
                    LD(x2, j + 48) AD(x3)
                    j += 32;
                }
                AD(x0) AD(x1) AD(x2)
            }
        }
    } else if (V == 5 || V == 6) {
        const int row = lane >> 4, l16 = lane & 15;
        const double *mine = lx + row * (T / 4);            // the row's list: T / 4 elements
        for (int r = 0; r < reps; r++) {
            double e = mine[l16], en;
            for (int k = 0; k < T / 4; k += 16) {
                en = mine[((k + 16) & (T / 4 - 1)) + l16];
                const unsigned own = (unsigned)(__double2loint(e) >> 3) & 15u;       // V == 6: some owner lane
#define STEP(u) { double b; asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #u " row_mask:0xf bank_mask:0xf" : "=v"(b) : "v"(e)); \
                  if (V == 6) { unsigned ob; asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:" #u " row_mask:0xf bank_mask:0xf" : "=v"(ob) : "v"(own)); \
                                s += (ob == (unsigned)l16) ? b : 0.0; } else s += b; }
                STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
#undef STEP
                e = en;
            }
        }
    } else {
        for (int r = 0; r < reps; r++) {
            double v = lx[lane], vn;
            for (int k = 0; k < T; k += 64) {
                vn = lx[((k + 64) & (T - 1)) + lane];
                const int lo = __double2loint(v), hi = __double2hiint(v);
                if (V == 3 && lane != 0) { v = vn; continue; }
#pragma unroll
                for (int u = 0; u < 64; u++) {
                    const int a = __builtin_amdgcn_readlane(lo, u), b = __builtin_amdgcn_readlane(hi, u);
                    s += __hiloint2double(b, a);
                }
                v = vn;
            }
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = s; cyc[0] = t1 - t0; }
}
template <int V> void run(const double *x, double *out, long long *cyc)
{
    const int reps = 200;
    hipLaunchKernelGGL(k_walk<V>, dim3(1), dim3(64), 0, 0, x, reps, out, cyc);
    hipDeviceSynchronize();
    long long c; double r; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&r, out, 8, hipMemcpyDeviceToHost);
    const double n = (double)reps * (V <= 1 || V == 4 ? (T - 8) : T);
    if (V >= 5) printf("variant %d: %.2f cycles per step of a row = %.2f cycles per element with four rows (sum %.17g)\n", V, (double)c / (n / 4), (double)c / n, r);
    else printf("variant %d: %.2f cycles per element (sum %.17g)\n", V, (double)c / n, r);
}
int main()
{
    double *x, *out; long long *cyc;
    hipMalloc(&x, T * 8); hipMalloc(&out, 64); hipMalloc(&cyc, 64);
    double h[T]; for (int i = 0; i < T; i++) h[i] = 1e-3 * (i % 17) - 7e-3 + 1e-9 * i;
    hipMemcpy(x, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(x, out, cyc); run<1>(x, out, cyc); run<2>(x, out, cyc); run<3>(x, out, cyc); run<5>(x, out, cyc); run<6>(x, out, cyc);
    return 0;
}
