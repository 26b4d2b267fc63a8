// Write rate of the match kernel's store pattern against wider and fully contiguous ones:
// 32 images x 160 rows x 640 anchor columns, 4 B + 16 B per anchor (65.5 MB), nontemporal stores,
// nothing else in the kernel.  Per pattern: the kernel alone (events riding on the launch) and
// four launches in flight on four streams (wall time per launch).
//   A  tile 32 columns x 160 rows; a wave stores two rows x 32 columns per instruction
//   B  tile 64 columns x 80 rows; a wave stores one row x 64 columns per instruction
//   C  tile 128 columns x 40 rows; a wave stores one row x 64 columns, two waves side by side
//   L  a wave stores 64 consecutive anchors per instruction, a workgroup a contiguous range
//   P  pattern A behind a prologue like the match kernel's: one round trip to memory, two
//      barriers and about a microsecond of arithmetic before the first store
//   Q  pattern A at the match kernel's occupancy (five waves per SIMD) without a prologue
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

constexpr int kRows = 160, kCols = 640, kBatch = 32, kPerImage = kRows * kCols;
typedef float vf4 __attribute__((ext_vector_type(4)));

__device__ inline void put(float *conf, vf4 *locs, long long a, float v)
{
    __builtin_nontemporal_store(v, conf + a);
    vf4 t = {v, v + 1.f, v + 2.f, v + 3.f};
    __builtin_nontemporal_store(t, locs + a);
}

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float *conf, vf4 *locs, float v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long img = (long long)blockIdx.y * kPerImage;
    __shared__ float s_x[256];
    __shared__ char s_pad[MODE == 5 ? 30000 : 1];      // five workgroups per CU
    if (MODE == 5 && v == 12345.f) s_pad[threadIdx.x] = 1;
    if (MODE == 4) {
        // what was stored last time comes back (a dependent round trip), then barriers and arithmetic
        float x = conf[img + (long long)blockIdx.x * 32 + threadIdx.x];
        s_x[threadIdx.x] = x;
        __syncthreads();
        float y = s_x[255 - threadIdx.x];
        for (int i = 0; i < 150; ++i) y = y / (x + 1.5f) + 0.25f;      // ~1 us of dependent divisions
        s_x[threadIdx.x] = y;
        __syncthreads();
        v += s_x[(threadIdx.x + 1) & 255] * 1e-30f;
    }
    if (MODE == 0 || MODE == 4 || MODE == 5) {                // A: blockIdx.x = column tile of 32
        const int c = blockIdx.x * 32 + (lane & 31), half = lane >> 5;
        for (int g = wave; g < kRows / 8; g += 4)
#pragma unroll
            for (int i = 0; i < 4; ++i) put(conf, locs, img + (long long)(g * 8 + half + 2 * i) * kCols + c, v);
    } else if (MODE == 1) {         // B: blockIdx.x = (column tile of 64, row half)
        const int tx = blockIdx.x >> 1, ty = blockIdx.x & 1;
        const int c = tx * 64 + lane;
        for (int g = wave; g < 80 / 4; g += 4)
#pragma unroll
            for (int i = 0; i < 4; ++i) put(conf, locs, img + (long long)(ty * 80 + g * 4 + i) * kCols + c, v);
    } else if (MODE == 2) {         // C: blockIdx.x = (column tile of 128, row quarter)
        const int tx = blockIdx.x >> 2, ty = blockIdx.x & 3;
        const int c = tx * 128 + (wave & 1) * 64 + lane;
        for (int g = wave >> 1; g < 40 / 4; g += 2)
#pragma unroll
            for (int i = 0; i < 4; ++i) put(conf, locs, img + (long long)(ty * 40 + g * 4 + i) * kCols + c, v);
    } else {                        // L: blockIdx.x = contiguous range of 5120 anchors
        const long long base = img + (long long)blockIdx.x * 5120 + wave * 1280;
        for (int g = 0; g < 5; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) put(conf, locs, base + (g * 4 + i) * 64 + lane, v);
    }
}

template <int MODE>
void run(const char *name, hipStream_t *s, float **conf, vf4 **locs)
{
    const dim3 grid(20, kBatch);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(store_kernel<MODE>, grid, dim3(256), 0, s[0], conf[0], locs[0], 1.f);
    hipStreamSynchronize(s[0]);
    float sum = 0.f;
    for (int i = 0; i < 30; ++i) {
        hipExtLaunchKernelGGL(store_kernel<MODE>, grid, dim3(256), 0, s[0], e0, e1, 0, conf[0], locs[0], 1.f);
        hipStreamSynchronize(s[0]);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        sum += ms;
    }
    const int n = 400;
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(store_kernel<MODE>, grid, dim3(256), 0, s[i & 3], conf[i & 3], locs[i & 3], 1.f);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(store_kernel<MODE>, grid, dim3(256), 0, s[i & 3], conf[i & 3], locs[i & 3], 1.f);
    hipDeviceSynchronize();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    const double mb = (double)kBatch * kPerImage * 20 / 1e6;
    printf("%s: alone %.2f us (%.2f TB/s), four in flight %.2f us per launch (%.2f TB/s)\n", name, sum / 30 * 1e3,
           mb / (sum / 30 * 1e3), us, mb / us);
}

int main()
{
    hipStream_t s[4];
    float *conf[4];
    vf4 *locs[4];
    for (int i = 0; i < 4; ++i) {
        hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
        hipMalloc(&conf[i], (size_t)kBatch * kPerImage * 4);
        hipMalloc(&locs[i], (size_t)kBatch * kPerImage * 16);
    }
    run<0>("A 2 rows x 32 columns", s, conf, locs);
    run<1>("B 1 row x 64 columns ", s, conf, locs);
    run<2>("C 128-column tiles   ", s, conf, locs);
    run<3>("L contiguous         ", s, conf, locs);
    run<4>("P A behind a prologue", s, conf, locs);
    run<5>("Q A, 5 workgroups/CU ", s, conf, locs);
    run<0>("A again              ", s, conf, locs);
    return 0;
}
