// What bounds a chain "streaming read of one batch, then a few short per-image kernels" when
// several batches are in flight on different streams: every kernel stamps its begin and end on
// the device's wall clock, so the overlap is seen without a profiler or events in the way.
//   read_kernel    streams a 78 MB buffer the way the select kernel does (six 1-KiB loads per
//                  wave), one of several buffers in rotation (more than the 256 MiB cache holds)
//   tail_kernel    G workgroups of T threads that each stay D microseconds (a stand-in for the
//                  top-k / mask / scan kernels: latency-bound, little memory traffic)
// Usage: chain_stamps STREAMS "G,T,D;G,T,D;..." [rounds] [read_wgs_per_cu] [read_threads] [read_lds_bytes] [read_mode]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned long long u64;

extern __shared__ unsigned char dyn_lds[];

// MODE 0: six 16-byte loads per lane and item (the whole rows); MODE d > 0: the face scores only,
// one dword per anchor at a 24-byte stride (four loads per lane and item), d items in flight.
template <int MODE>
__global__ __launch_bounds__(1024) void read_kernel(const float4 *__restrict__ src, int n_items, float thr,
                                                    int *__restrict__ out, u64 *stamp)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = wall_clock64();
    const int lane = threadIdx.x & 63;
    const int wpb = (int)blockDim.x >> 6;
    const int stride = (int)gridDim.x * wpb;
    int hits = 0;
    if (MODE == 0) {
        for (int item = blockIdx.x * wpb + (threadIdx.x >> 6); item < n_items; item += stride) {
            const float4 *p = src + (long long)item * 384 + lane;
            float4 v[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) v[k] = p[64 * k];
#pragma unroll
            for (int k = 0; k < 6; ++k) hits += (v[k].y > thr) + (v[k].w > thr);
        }
    } else {
        constexpr int D = MODE > 0 ? MODE : 1;
        const float *f = reinterpret_cast<const float *>(src);
        for (int item = blockIdx.x * wpb + (threadIdx.x >> 6); item < n_items; item += stride * D) {
            float v[D][4];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int it = min(item + d * stride, n_items - 1);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[d][k] = f[((long long)it * 256 + k * 64 + lane) * 6 + 1];
            }
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int k = 0; k < 4; ++k) hits += v[d][k] > thr;
        }
    }
    if (hits == 12345678) out[0] = hits + dyn_lds[threadIdx.x];
    if (threadIdx.x == 0) stamp[1] = wall_clock64();
}

// Ticket-ordered variant: the host numbers the read kernels in launch order; a workgroup waits
// (bounded) until the ticket being served is its kernel's, the last workgroup of a kernel to leave
// passes the turn on.  One small workgroup per CU, so that waiting kernels hold few wave slots.
__global__ __launch_bounds__(256) void read_ticket_kernel(const float4 *__restrict__ src, int n_items, float thr,
                                                          int *__restrict__ out, u64 *stamp, int *turn, int ticket)
{
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ticket && spins < 4000) {
            __builtin_amdgcn_s_sleep(16);
            ++spins;
        }
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = wall_clock64();
    const int lane = threadIdx.x & 63;
    const int wpb = (int)blockDim.x >> 6;
    const int stride = (int)gridDim.x * wpb;
    int hits = 0;
    const float *f = reinterpret_cast<const float *>(src);
    for (int item = blockIdx.x * wpb + (threadIdx.x >> 6); item < n_items; item += stride * 4) {
        float v[4][4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int it = min(item + d * stride, n_items - 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[d][k] = f[((long long)it * 256 + k * 64 + lane) * 6 + 1];
        }
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int k = 0; k < 4; ++k) hits += v[d][k] > thr;
    }
    if (hits == 12345678) out[0] = hits;
    __syncthreads();
    if (threadIdx.x == 0) {
        stamp[1] = wall_clock64();
        if (atomicAdd(turn + 64, 1) == (int)gridDim.x - 1) {
            __hip_atomic_store(turn + 64, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(turn, ticket + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void tail_kernel(int us, int *out, u64 *stamp)
{
    const u64 t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = t0;
    const u64 ticks = (u64)us * 100;          // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (us < 0) out[0] = 1;
    if (threadIdx.x == 0) stamp[1] = wall_clock64();
}

struct Tail {
    int g, t, d;
};

int main(int argc, char **argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 4;
    std::vector<Tail> tails;
    if (argc > 2) {
        char *spec = strdup(argv[2]);
        for (char *tok = strtok(spec, ";"); tok; tok = strtok(nullptr, ";")) {
            Tail t;
            if (sscanf(tok, "%d,%d,%d", &t.g, &t.t, &t.d) == 3) tails.push_back(t);
        }
    }
    const int rounds = argc > 3 ? atoi(argv[3]) : 300;
    const int read_wgs_per_cu = argc > 4 ? atoi(argv[4]) : 0;      // 0: one item per wave
    const int read_threads = argc > 5 ? atoi(argv[5]) : 256;
    const int read_lds = argc > 6 ? atoi(argv[6]) : 0;             // dynamic LDS bytes per read workgroup
    const int read_mode = argc > 7 ? atoi(argv[7]) : 0;
    void (*rk)(const float4 *, int, float, int *, u64 *) =
        read_mode == 0 ? read_kernel<0> : read_mode == 1 ? read_kernel<1> : read_mode == 2 ? read_kernel<2>
        : read_mode == 3 ? read_kernel<3> : read_mode == 4 ? read_kernel<4> : read_kernel<6>;
    const int n_items = 32 * 399;                                  // batch 32, 102 000 anchors
    const size_t bytes = (size_t)n_items * 384 * 16;
    const int NB = std::max(S, 4);
    std::vector<float4 *> buf(NB);
    for (auto &b : buf) {
        hipMalloc(&b, bytes);
        hipMemset(b, 0, bytes);
    }
    int *d_out;
    hipMalloc(&d_out, 64);
    const int K = 1 + (int)tails.size();
    u64 *d_stamp;
    const size_t n_stamps = (size_t)rounds * S * K * 2;
    hipMalloc(&d_stamp, n_stamps * sizeof(u64));
    hipMemset(d_stamp, 0, n_stamps * sizeof(u64));
    std::vector<hipStream_t> st(S);
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    for (auto &s : st) hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi);
    const int wpb = read_threads / 64;
    const int read_grid = read_wgs_per_cu ? 256 * read_wgs_per_cu : (n_items + wpb - 1) / wpb;
    if (read_lds > 65536)
        hipFuncSetAttribute((const void *)rk, hipFuncAttributeMaxDynamicSharedMemorySize, read_lds);

    // [read_mode >= 10]: every read kernel goes to ONE shared stream (first come, first served at
    // full bandwidth) and the tails of batch s wait for it through an event; the shared stream in
    // turn waits for an event recorded on stream s before the call (the batch's inputs)
    const bool fifo = getenv("CHAIN_FIFO") != nullptr;
    hipStream_t s_read;
    hipStreamCreateWithPriority(&s_read, hipStreamNonBlocking, hi);
    std::vector<hipEvent_t> e_in(S), e_rd(S);
    for (int s = 0; s < S; ++s) {
        hipEventCreateWithFlags(&e_in[s], hipEventDisableTiming);
        hipEventCreateWithFlags(&e_rd[s], hipEventDisableTiming);
    }
    const bool ticketed = getenv("CHAIN_TICKET") != nullptr;
    int *d_turn;
    hipMalloc(&d_turn, 1024);
    hipMemset(d_turn, 0, 1024);
    int next_ticket = 0;
    auto run = [&](int r0, int r1) {
        for (int r = r0; r < r1; ++r)
            for (int s = 0; s < S; ++s) {
                u64 *sp = d_stamp + ((size_t)(r * S + s) * K) * 2;
                if (ticketed) {
                    hipLaunchKernelGGL(read_ticket_kernel, dim3(256 * std::max(read_wgs_per_cu, 1)), dim3(read_threads), 0, st[s],
                                       buf[(r * S + s) % NB], n_items, 2.0f, d_out, sp, d_turn, next_ticket++);
                    for (int k = 0; k < (int)tails.size(); ++k)
                        hipLaunchKernelGGL(tail_kernel, dim3(tails[k].g), dim3(tails[k].t), 0, st[s], tails[k].d, d_out,
                                           sp + 2 * (k + 1));
                    continue;
                }
                if (fifo) {
                    hipEventRecord(e_in[s], st[s]);
                    hipStreamWaitEvent(s_read, e_in[s], 0);
                    hipLaunchKernelGGL(rk, dim3(read_grid), dim3(read_threads), read_lds, s_read, buf[(r * S + s) % NB],
                                       n_items, 2.0f, d_out, sp);
                    hipEventRecord(e_rd[s], s_read);
                    hipStreamWaitEvent(st[s], e_rd[s], 0);
                    for (int k = 0; k < (int)tails.size(); ++k)
                        hipLaunchKernelGGL(tail_kernel, dim3(tails[k].g), dim3(tails[k].t), 0, st[s], tails[k].d, d_out,
                                           sp + 2 * (k + 1));
                    continue;
                }
                hipLaunchKernelGGL(rk, dim3(read_grid), dim3(read_threads), read_lds, st[s], buf[(r * S + s) % NB],
                                   n_items, 2.0f, d_out, sp);
                for (int k = 0; k < (int)tails.size(); ++k)
                    hipLaunchKernelGGL(tail_kernel, dim3(tails[k].g), dim3(tails[k].t), 0, st[s], tails[k].d, d_out,
                                       sp + 2 * (k + 1));
            }
    };
    run(0, 20);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    run(0, rounds);
    const auto t_enq = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    const auto t1 = std::chrono::steady_clock::now();
    const double wall = std::chrono::duration<double, std::micro>(t1 - t0).count() / (rounds * S);
    const double enq = std::chrono::duration<double, std::micro>(t_enq - t0).count() / (rounds * S);

    std::vector<u64> h(n_stamps);
    hipMemcpy(h.data(), d_stamp, n_stamps * sizeof(u64), hipMemcpyDeviceToHost);
    // steady part: the middle half of the rounds
    const int ra = rounds / 4, rb = 3 * rounds / 4;
    std::vector<double> dur(K, 0.0), gap(K, 0.0);
    double chain = 0.0;
    int n = 0;
    std::vector<std::pair<u64, int>> ev;
    for (int r = ra; r < rb; ++r)
        for (int s = 0; s < S; ++s) {
            const u64 *sp = h.data() + ((size_t)(r * S + s) * K) * 2;
            for (int k = 0; k < K; ++k) {
                dur[k] += (double)(sp[2 * k + 1] - sp[2 * k]) * 0.01;
                if (k > 0) gap[k] += (double)(sp[2 * k] - sp[2 * k - 1]) * 0.01;
            }
            chain += (double)(sp[2 * K - 1] - sp[0]) * 0.01;
            ev.push_back({sp[0], 1});
            ev.push_back({sp[1], -1});
            ++n;
        }
    std::sort(ev.begin(), ev.end());
    double busy[4] = {0, 0, 0, 0};
    int level = 0;
    u64 prev = ev.front().first;
    for (auto &e : ev) {
        busy[std::min(level, 3)] += (double)(e.first - prev);
        prev = e.first;
        level += e.second;
    }
    const double tot = busy[0] + busy[1] + busy[2] + busy[3];
    printf("read %d thr, %d B LDS, mode %d; ", read_threads, read_lds, read_mode);
    printf("streams %d tails %s read_grid %d: wall %.2f us/batch (%.2fe6 img/s, %.0f GB/s) host enqueue %.2f us/batch\n", S,
           argc > 2 ? argv[2] : "-", read_grid, wall, 32.0 / wall, bytes / wall * 1e-3, enq);
    printf("   read %.2f us", dur[0] / n);
    for (int k = 1; k < K; ++k) printf(" | gap %.2f tail%d %.2f", gap[k] / n, k, dur[k] / n);
    printf(" | chain %.2f\n   reads running 0: %.1f%%  1: %.1f%%  2: %.1f%%  3+: %.1f%%\n", chain / n, 100 * busy[0] / tot,
           100 * busy[1] / tot, 100 * busy[2] / tot, 100 * busy[3] / tot);
    return 0;
}
