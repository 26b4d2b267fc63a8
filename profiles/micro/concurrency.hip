// How kernels of different streams share the chip: what a per-image kernel (few large
// workgroups, latency-bound) costs next to a streaming kernel that fills every CU, by workgroup
// size of the small kernel and by occupancy of the streaming one.
//   spin_kernel    G workgroups of T threads, each waits D microseconds on the wall clock
//   fill_kernel    streaming 16-byte stores of `bytes`, one 4-KiB chunk per wave iteration;
//                  persistent = grid-stride loop on 256 * wgs_per_cu workgroups
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(int us, int *out)
{
    const long long t0 = wall_clock64();
    const long long ticks = (long long)us * 100;          // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (us < 0) out[0] = 1;
}

__global__ __launch_bounds__(256) void fill_kernel(float4 *dst, long long n4, int persistent)
{
    const long long stride = persistent ? (long long)gridDim.x * 256 : n4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
        dst[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    const int S = 8;
    hipStream_t st[S];
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    for (int i = 0; i < S; ++i) hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, hi);
    int *d_out;
    hipMalloc(&d_out, 64);
    const long long bytes = 65ll << 20;
    float4 *buf[S];
    for (int i = 0; i < S; ++i) hipMalloc(&buf[i], bytes);
    const long long n4 = bytes / 16;

    // 1. spin kernels alone: S streams x K kernels of 32 x 1024 threads, 20 us each
    for (int threads : {1024, 256}) {
        for (int ns : {1, 2, 4, 8}) {
            const int K = 50;
            for (int w = 0; w < 2; ++w) {
                const double t0 = now_us();
                for (int k = 0; k < K; ++k)
                    for (int s = 0; s < ns; ++s)
                        hipLaunchKernelGGL(spin_kernel, dim3(32 * 1024 / threads), dim3(threads), 0, st[s], 20, d_out);
                hipDeviceSynchronize();
                if (w == 1)
                    printf("spin 20us, %4d thr/wg, %d streams: %.2f us per kernel (ideal %.2f)\n", threads, ns,
                           (now_us() - t0) / (K * ns), 20.0 / ns);
            }
        }
    }
    // 2. fill alone, by shape
    for (int persistent : {0, 4, 8}) {
        const int grid = persistent ? 256 * persistent : (int)((n4 + 255) / 256);
        for (int ns : {1, 4}) {
            const int K = 50;
            for (int w = 0; w < 2; ++w) {
                const double t0 = now_us();
                for (int k = 0; k < K; ++k)
                    for (int s = 0; s < ns; ++s)
                        hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(256), 0, st[s], buf[s], n4, persistent);
                hipDeviceSynchronize();
                if (w == 1)
                    printf("fill 65MiB, persistent %d wg/cu, %d streams: %.2f us per kernel (%.0f GB/s)\n", persistent, ns,
                           (now_us() - t0) / (K * ns), bytes / ((now_us() - t0) / (K * ns)) * 1e-3);
            }
        }
    }
    // 3. chains fill -> spin on every stream (the shape of select -> per-image kernel)
    for (int threads : {1024, 256}) {
        for (int persistent : {0, 4}) {
            const int grid = persistent ? 256 * persistent : (int)((n4 + 255) / 256);
            for (int ns : {1, 4, 8}) {
                const int K = 50;
                for (int w = 0; w < 2; ++w) {
                    const double t0 = now_us();
                    for (int k = 0; k < K; ++k)
                        for (int s = 0; s < ns; ++s) {
                            hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(256), 0, st[s], buf[s], n4, persistent);
                            hipLaunchKernelGGL(spin_kernel, dim3(32 * 1024 / threads), dim3(threads), 0, st[s], 20, d_out);
                        }
                    hipDeviceSynchronize();
                    if (w == 1)
                        printf("chain fill(persistent %d) -> spin 20us (%4d thr/wg), %d streams: %.2f us per pair\n",
                               persistent, threads, ns, (now_us() - t0) / (K * ns));
                }
            }
        }
    }
    // 4. chain of three spins (dependent small kernels) on every stream
    for (int ns : {1, 4, 8}) {
        const int K = 50;
        for (int w = 0; w < 2; ++w) {
            const double t0 = now_us();
            for (int k = 0; k < K; ++k)
                for (int s = 0; s < ns; ++s)
                    for (int q = 0; q < 3; ++q)
                        hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(1024), 0, st[s], 10, d_out);
            hipDeviceSynchronize();
            if (w == 1) printf("chain of 3 spins 10us, %d streams: %.2f us per chain (ideal %.2f)\n", ns, (now_us() - t0) / (K * ns), 30.0 / ns);
        }
    }
    return 0;
}
