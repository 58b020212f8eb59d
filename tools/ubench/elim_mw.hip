// Micro-benchmark ("next" 2): Gaussian elimination of a symmetric positive definite matrix of 64 * NW rows by NW cooperating
// waves of ONE workgroup, lane = row, the row's N entries in registers.  The pivot row cannot be broadcast with v_readlane across waves; the
// matrix stays symmetric under elimination, so column k of the trailing block IS row k: at step k every lane writes its entry of column k
// to LDS (one ds_write per wave), a barrier, and every lane reads the pivot row back with broadcast reads (all lanes the same address).
// Two LDS buffers alternate, so one barrier per pivot suffices.
//   NW = 1: against the register-resident v_readlane version of the product kernel (as_solve; elim.hip): what the LDS round trip costs;
//   NW = 2: 128 rows, 256 VGPRs per lane for the matrix.  (Three waves would need 384 VGPRs per lane for 192 rows, four 512 for the matrix alone.)
//   k_solve: the whole solve A f = -b on NW waves (right-hand side carried along, back substitution through LDS), checked on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 elim_mw.hip -o elim_mw && ./elim_mw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
template <int B, int E, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
__device__ __forceinline__ double bcast(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// reference: one wave, v_readlane broadcasts (the product kernel's scheme)
__global__ void __launch_bounds__(64) k_readlane(double* out, long long* cyc, double a, int reps) {
    constexpr int N = 64;
    const int lane = threadIdx.x;
    double W[N];
    static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; W[j] = (j == lane ? 40.0 : 0.0) + a * (((lane * 7 + j * 13) % 17) + ((j * 7 + lane * 13) % 17)); });
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        static_for<0, N>([&](auto kc) __attribute__((always_inline)) {
            constexpr int kk = decltype(kc)::value;
            const double pk = bcast(W[kk], kk);
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const double l = ln > kk ? W[kk] / pk : 0.0;
            static_for<kk + 1, N>([&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; W[j] = fma(-l, bcast(W[j], kk), W[j]); });
        });
        W[0] += 1e-9;
    }
    long long t1 = clock64();
    double s = 0;
    static_for<0, N>([&](auto jc) { s += W[decltype(jc)::value]; });
    out[lane] = s;
    if (lane == 0) cyc[0] = t1 - t0;
}
template <int NW> __global__ void __launch_bounds__(64 * NW) k_lds(double* out, long long* cyc, double a, int reps) {
    constexpr int N = 64 * NW;
    __shared__ double col[2][N];
    const int row = threadIdx.x;
    double W[N];
    static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; W[j] = (j == row ? 40.0 * NW : 0.0) + a * (((row * 7 + j * 13) % 17) + ((j * 7 + row * 13) % 17)); });
    __syncthreads();
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        static_for<0, N>([&](auto kc) __attribute__((always_inline)) {
            constexpr int kk = decltype(kc)::value;
            double* c = col[kk & 1];
            int rw = row;
            asm volatile("" : "+v"(rw));  // opaque per pivot: keeps the 64 * NW (row > k) masks and LDS addresses from being hoisted out of the loop (and spilled)
            c[rw] = W[kk];  // column kk of the trailing block = row kk (symmetric)
            __syncthreads();
            const double pk = c[kk];
            const double l = rw > kk ? W[kk] / pk : 0.0;
            // chunks of 8 columns: eight broadcast reads, eight FMAs
            static_for<(kk + 1) / 8, N / 8>([&](auto cc) __attribute__((always_inline)) {
                constexpr int ch = decltype(cc)::value;
                constexpr int j0 = 8 * ch > kk + 1 ? 8 * ch : kk + 1;
                double y8[8];
                static_for<j0, 8 * ch + 8>([&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; y8[j - 8 * ch] = c[j]; });
                static_for<j0, 8 * ch + 8>([&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; W[j] = fma(-l, y8[j - 8 * ch], W[j]); asm volatile("" : "+v"(W[j])); });
                // (the empty asm statements pin every update to its pivot and every read to its chunk: left alone the compiler defers the updates of
                //  several pivots into one FMA chain per entry and reads whole pivot rows up front -- 2 000+ spilled registers)
                asm volatile("" ::: "memory");
            });
        });
        W[0] += 1e-9;
        __syncthreads();
    }
    long long t1 = clock64();
    double s = 0;
    static_for<0, N>([&](auto jc) { s += W[decltype(jc)::value]; });
    out[row] = s;
    if (row == 0) cyc[0] = t1 - t0;
}
// Full solve A f = -b of 64 * NW rows on NW waves: the elimination above with the right-hand side carried along (its pivot entry goes through
// LDS with the pivot row), then the back substitution column by column -- lane k divides, x_k goes through LDS, every lane above subtracts
// its entry of column k.  One barrier per pivot in each phase.  The host checks max |A f + b|.
template <int NW> __global__ void __launch_bounds__(64 * NW) k_solve(double* out, long long* cyc, double a, int reps) {
    constexpr int N = 64 * NW;
    __shared__ double col[2][N + 2];
    const int row = threadIdx.x;
    double W[N], f = 0.0;
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; W[j] = (j == row ? 40.0 * NW : 0.0) + a * (((row * 7 + j * 13) % 17) + ((j * 7 + row * 13) % 17)); });
        double c = -(1.0 + 0.01 * row);  // -b
        static_for<0, N>([&](auto kc) __attribute__((always_inline)) {
            constexpr int kk = decltype(kc)::value;
            double* cb = col[kk & 1];
            int rw = row;
            asm volatile("" : "+v"(rw));
            cb[rw] = W[kk];
            if (rw == kk) cb[N] = c;
            __syncthreads();
            const double pk = cb[kk], ck = cb[N];
            const double l = rw > kk ? W[kk] / pk : 0.0;
            c = fma(-l, ck, c);
            static_for<(kk + 1) / 8, N / 8>([&](auto cc) __attribute__((always_inline)) {
                constexpr int ch = decltype(cc)::value;
                constexpr int j0 = 8 * ch > kk + 1 ? 8 * ch : kk + 1;
                double y8[8];
                static_for<j0, 8 * ch + 8>([&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; y8[j - 8 * ch] = cb[j]; });
                static_for<j0, 8 * ch + 8>([&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; W[j] = fma(-l, y8[j - 8 * ch], W[j]); asm volatile("" : "+v"(W[j])); });
                asm volatile("" ::: "memory");
            });
        });
        __syncthreads();
        // back substitution: acc_i = c_i - sum_{k > i} U[i][k] x_k, x_k = acc_k / U[k][k]
        double acc = c;
        static_for<0, N>([&](auto kc) __attribute__((always_inline)) {
            constexpr int kk = N - 1 - decltype(kc)::value;
            double* cb = col[kk & 1];
            int rw = row;
            asm volatile("" : "+v"(rw));
            if (rw == kk) { f = acc / W[kk]; cb[N] = f; }
            __syncthreads();
            const double xk = cb[N];
            if (rw < kk) acc = fma(-W[kk], xk, acc);
            asm volatile("" : "+v"(acc));
        });
        __syncthreads();
    }
    long long t1 = clock64();
    out[row] = f;
    if (row == 0) cyc[0] = t1 - t0;
}
template <int NW> double residual(const double* f, double a) {
    constexpr int N = 64 * NW;
    double worst = 0;
    for (int i = 0; i < N; i++) {
        double y = 1.0 + 0.01 * i;
        for (int j = 0; j < N; j++) y += ((j == i ? 40.0 * NW : 0.0) + a * (((i * 7 + j * 13) % 17) + ((j * 7 + i * 13) % 17))) * f[j];
        worst = y > worst ? y : (-y > worst ? -y : worst);
    }
    return worst;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 8);
    const int reps = 10;
    auto report = [&](const char* nm, int n) {
        hipDeviceSynchronize();
        long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        double host[256]; hipMemcpy(host, out, n * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < n; i++) s += host[i];
        printf("%-64s %9.0f cycles / elimination, %5.2f cycles / element update (checksum %.6e)\n", nm, h / (double)reps, h / (double)reps / (n * (double)(n - 1) / 2), s);
    };
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_readlane, dim3(1), dim3(64), 0, 0, out, cyc, 1e-3, reps);
    report("64 rows, 1 wave, v_readlane broadcasts (product scheme)", 64);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_lds<1>, dim3(1), dim3(64), 0, 0, out, cyc, 1e-3, reps);
    report("64 rows, 1 wave, pivot row through LDS", 64);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_lds<2>, dim3(1), dim3(128), 0, 0, out, cyc, 1e-3, reps);
    report("128 rows, 2 waves, pivot row through LDS", 128);
    {
        double host[256];
        for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_solve<1>, dim3(1), dim3(64), 0, 0, out, cyc, 1e-3, reps);
        hipDeviceSynchronize();
        long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(host, out, 64 * 8, hipMemcpyDeviceToHost);
        printf("%-64s %9.0f cycles / solve (build + elimination + back substitution), max |A f + b| = %.2e\n", "64 rows, 1 wave, full solve through LDS", h / (double)reps, residual<1>(host, 1e-3));
        for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_solve<2>, dim3(1), dim3(128), 0, 0, out, cyc, 1e-3, reps);
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(host, out, 128 * 8, hipMemcpyDeviceToHost);
        printf("%-64s %9.0f cycles / solve (build + elimination + back substitution), max |A f + b| = %.2e\n", "128 rows, 2 waves, full solve through LDS", h / (double)reps, residual<2>(host, 1e-3));
    }
    return 0;
}
