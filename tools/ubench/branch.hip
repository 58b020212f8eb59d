// Cost of wave-uniform branches with one wave per SIMD (gfx950): 64 unrolled `if (bit i of a scalar mask) { one FMA }` tests,
// all bits set (every branch falls through into the body), none set (every branch jumps over it), alternating.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 branch.hip -o branch && ./branch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
template <int B, int E, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
template <bool OPAQUE> __global__ void k(double* out, long long* cyc, unsigned long long mask, double a, int reps) {
    double x = a + threadIdx.x, y = a;
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        unsigned long long m = mask;
        if (OPAQUE) asm volatile("" : "+s"(m));
        static_for<0, 64>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if ((m >> i) & 1ull) { x = fma(x, a, y); asm volatile("" : "+v"(x)); }
        });
    }
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
    const unsigned long long masks[] = {~0ull, 0ull, 0x5555555555555555ull, 0x00000000ffffffffull};
    const char* nm[] = {"all taken into the body (fall through)", "all skipped (branch taken)", "alternating", "first 32 in, last 32 skipped"};
    for (int v = 0; v < 4; v++) {
        for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k<true>, dim3(1), dim3(64), 0, 0, out, cyc, masks[v], 1.0000001, 50);
        (void)hipDeviceSynchronize();
        long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-44s %7.1f cycles per test (incl. %d dependent FMAs of ~6 cycles per 64 tests)\n", nm[v], h / 50.0 / 64.0, __builtin_popcountll(masks[v]));
    }
    return 0;
}
