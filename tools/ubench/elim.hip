// Micro-benchmark for the register-resident Gaussian elimination of the active-set contact solver (uhc_physics.hip, as_solve):
// lane = row, W[j] = column j in registers, step k broadcasts lane k's entries with two v_readlane each.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 elim.hip -o elim && ./elim
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
template <int B, int E, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
__device__ __forceinline__ double bcast(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <int N, bool SKIP, bool RCP, int NC = N, bool CHUNK = true> __global__ void k(double* out, long long* cyc, double a, int reps, unsigned long long mask, int nefc) {
    const int lane = threadIdx.x;
    double W[N];
    static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; W[j] = (j == lane ? 4.0 : 0.0) + a * ((lane * 7 + j * 13) % 17); });
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        static_for<0, N>([&](auto kc) __attribute__((always_inline)) {
            constexpr int kk = decltype(kc)::value;
            if (!SKIP || ((mask >> kk) & 1ull)) {
                const double pk = bcast(W[kk], kk);
                const double pinv = RCP ? __builtin_amdgcn_rcp(pk) : 1.0 / pk;
                const double l = lane > kk ? W[kk] * pinv : 0.0;
                static_for<(kk + 1) / 8, NC / 8>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int ch = decltype(cc)::value;
                    if (!SKIP || !CHUNK || 8 * ch < nefc) {
                        static_for<(8 * ch > kk + 1 ? 8 * ch : kk + 1), 8 * ch + 8>([&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value;
                            W[j] = fma(-l, bcast(W[j], kk), W[j]);
                        });
                    }
                });
            }
        });
        W[0] += 1e-9;
    }
    long long t1 = clock64();
    double s = 0;
    static_for<0, N>([&](auto jc) { s += W[decltype(jc)::value]; });
    out[lane] = s;
    if (lane == 0) cyc[0] = t1 - t0;
}
template <int N, bool SKIP, bool RCP, int NC = N, bool CHUNK = true> void run(const char* nm, double* out, long long* cyc, unsigned long long mask, int nefc, double elems) {
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL((k<N, SKIP, RCP, NC, CHUNK>), dim3(1), dim3(64), 0, 0, out, cyc, 1e-3, 20, mask, nefc);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-52s %9.0f cycles / elimination, %6.1f cycles / element\n", nm, h / 20.0, h / 20.0 / elems);
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    run<32, false, false>("N=32 dense (496 elements, ~12 KB code)", out, cyc, ~0ull, 32, 496);
    run<64, false, false>("N=64 dense (2016 elements, ~50 KB code)", out, cyc, ~0ull, 64, 2016);
    run<64, false, true>("N=64 dense, v_rcp_f64 pivot", out, cyc, ~0ull, 64, 2016);
    // typical solve: 40 rows, 27 of them free (every third skipped)
    unsigned long long m = 0; int cnt = 0; double el = 0;
    for (int i = 0; i < 40; i++) if (i % 3 != 2) { m |= 1ull << i; cnt++; el += 39 - i; }
    run<64, true, false>("N=64 code, nefc=40, 27 active steps (skips)", out, cyc, m, 40, el);
    run<64, true, true>("same, v_rcp_f64 pivot", out, cyc, m, 40, el);
    run<64, true, true, 64, false>("rcp, step skips only (all 64 columns)", out, cyc, m, 40, el);
    run<64, true, true, 40, false>("rcp, step skips, columns < 40 by template", out, cyc, m, 40, el);
    run<64, false, true, 40, false>("rcp, no skips at all, columns < 40 by template", out, cyc, ~0ull, 40, 40 * 39 / 2);
    return 0;
}
