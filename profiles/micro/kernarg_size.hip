// Execution time of a kernel that returns at once, against the size of its by-value argument
// (events riding on the launch: the kernel's own begin/end timestamps) and the wall time per
// launch of a back-to-back chain.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

template <int N>
struct Args {
    int v[N];
};

template <int N>
__global__ void empty_kernel(const Args<N> a, int *out)
{
    if (a.v[0] == 12345 && threadIdx.x == 0) out[0] = a.v[N - 1];
}

template <int N>
void run(hipStream_t s, int *d_out, int blocks)
{
    Args<N> a = {};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_kernel<N>, dim3(blocks), dim3(256), 0, s, a, d_out);
    hipStreamSynchronize(s);
    float sum = 0.f;
    for (int i = 0; i < 50; ++i) {
        hipExtLaunchKernelGGL(empty_kernel<N>, dim3(blocks), dim3(256), 0, s, e0, e1, 0, a, d_out);
        hipStreamSynchronize(s);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        sum += ms;
    }
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(empty_kernel<N>, dim3(blocks), dim3(256), 0, s, a, d_out);
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 500;
    printf("args %5zu B, %5d blocks: kernel %.2f us, chain %.2f us per launch\n", sizeof(a), blocks, sum / 50 * 1e3, us);
}

int main()
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int *d_out;
    hipMalloc(&d_out, 64);
    for (int blocks : {32, 3400}) {
        run<16>(s, d_out, blocks);
        run<128>(s, d_out, blocks);
        run<576>(s, d_out, blocks);
        run<1000>(s, d_out, blocks);
    }
    return 0;
}
