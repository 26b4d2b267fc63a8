// Does hipExtAnyOrderLaunch let two kernels of one stream overlap on this device?  Two kernels
// that each wait 50 us on the wall clock, the second launched with the flag: 100 us means no.
//   hipcc --offload-arch=gfx950 -O2 -o any_order any_order.hip && ./any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void wait_us(unsigned long long ticks, unsigned long long *out)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(10);
    if (out) out[blockIdx.x] = wall_clock64();
}

int main()
{
    hipStream_t s;
    if (hipStreamCreate(&s) != hipSuccess) return 1;
    unsigned long long *d;
    if (hipMalloc(&d, 64 * 8) != hipSuccess) return 1;
    for (int flags = 0; flags < 2; ++flags) {
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            hipExtLaunchKernelGGL(wait_us, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0u, 5000ull, d);
            hipExtLaunchKernelGGL(wait_us, dim3(1), dim3(64), 0, s, nullptr, nullptr, (unsigned)flags, 5000ull, d + 1);
            (void)hipStreamSynchronize(s);
            auto t1 = std::chrono::steady_clock::now();
            printf("flags %d: two 50 us kernels on one stream took %.1f us\n", flags,
                   std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
    }
    return 0;
}
