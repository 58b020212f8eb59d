// Micro-benchmarks of the dependent-issue latencies that bound the one-wave-per-SIMD physics kernel
// (f64 VALU chain, v_readlane broadcast chain, LDS round trip).  hipcc --offload-arch=gfx950 -O3 lat.hip -o lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__global__ void k(double* out, long long* cyc, double a, double b) {
    __shared__ double sm[256];
    const int lane = threadIdx.x;
    sm[lane] = a * lane; sm[lane + 64] = b; sm[lane + 128] = lane; sm[lane + 192] = 0;
    __syncthreads();
    double x = a + lane, y = b, z = a * 2, w = b * 3;
    long long t0, t1;
    // 0: dependent fma chain
    t0 = clock64();
#pragma unroll 64
    for (int i = 0; i < N; i++) x = fma(x, a, b);
    t1 = clock64(); if (lane == 0) cyc[0] = t1 - t0;
    // 1: 4 independent fma chains
    t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; i++) { x = fma(x, a, b); y = fma(y, a, b); z = fma(z, a, b); w = fma(w, a, b); }
    t1 = clock64(); if (lane == 0) cyc[1] = t1 - t0;
    // 2: max -> sub -> readlane x2 -> fma (PGS row chain)
    double f = b, u = x;
    t0 = clock64();
#pragma unroll 64
    for (int i = 0; i < N; i++) {
        const double fn = fmax(u, 0.0);
        const double d = fn - f;
        const int lo = __builtin_amdgcn_readlane(__double2loint(d), 7), hi = __builtin_amdgcn_readlane(__double2hiint(d), 7);
        u = fma(-__hiloint2double(hi, lo), y, u);
    }
    t1 = clock64(); if (lane == 0) cyc[2] = t1 - t0;
    // 3: max(w, -f) -> readlane x2 -> fma (3-hop variant)
    t0 = clock64();
#pragma unroll 64
    for (int i = 0; i < N; i++) {
        const double d = fmax(u, -f);
        const int lo = __builtin_amdgcn_readlane(__double2loint(d), 7), hi = __builtin_amdgcn_readlane(__double2hiint(d), 7);
        u = fma(-__hiloint2double(hi, lo), y, u);
    }
    t1 = clock64(); if (lane == 0) cyc[3] = t1 - t0;
    // 4: dependent LDS read chain (address from data)
    int idx = lane;
    t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; i++) idx = (int)sm[128 + (idx & 63)];
    t1 = clock64(); if (lane == 0) cyc[4] = t1 - t0;
    // 5: LDS write -> barrier -> read round trip
    t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; i++) { sm[192 + lane] = z; __syncthreads(); z = sm[192 + ((lane + 1) & 63)] + 1.0; __syncthreads(); }
    t1 = clock64(); if (lane == 0) cyc[5] = t1 - t0;
    // 6: dependent add chain (f64)
    t0 = clock64();
#pragma unroll 64
    for (int i = 0; i < N; i++) w = w + a;
    t1 = clock64(); if (lane == 0) cyc[6] = t1 - t0;
    // 7: f64 division chain
    t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N / 8; i++) y = a / (y + 2.0);
    t1 = clock64(); if (lane == 0) cyc[7] = t1 - t0;
    // 8: DPP wave reduction chain (6 steps) as in wave_sum
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N / 8; i++) { double v = z; for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); z = v * 1e-3; }
    t1 = clock64(); if (lane == 0) cyc[8] = t1 - t0;
    // 9: sqrt chain
    t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N / 8; i++) x = sqrt(fabs(x) + 1.0);
    t1 = clock64(); if (lane == 0) cyc[9] = t1 - t0;
    out[lane] = x + y + z + w + u + idx;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16 * 8);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    long long h[16]; hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
    const char* nm[] = {"dep fma f64", "4 indep fma f64 (per fma)", "max-sub-readlane2-fma", "max-readlane2-fma", "dep LDS read", "LDS write+sync+read+sync",
                        "dep add f64", "div f64", "wave_sum via shfl_xor (6 steps)+mul", "sqrt f64"};
    const double div[] = {N, 4.0 * N, N, N, N, N, N, N / 8, N / 8, N / 8};
    for (int i = 0; i < 10; i++) printf("%-40s %8.1f cyc (wall_clock64 ticks)\n", nm[i], h[i] / div[i]);
    return 0;
}
