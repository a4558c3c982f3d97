// What does a hipGraph save between DEPENDENT launches on this GPU?  (measurement aid)
//     hipcc --offload-arch=gfx950 -O2 tools/graph_gap_ubench.hip -o /tmp/graph_gap && /tmp/graph_gap
// A kernel of the one-launch tick's shape (256 workgroups x 768 threads, busy for ~BUSY_US microseconds on wall_clock64, every
// launch reading what the one before wrote) is launched 4 096 times back to back on one stream, then as 64 captured graphs of 64
// kernel nodes each.  Per-launch time minus the busy time is what the launch boundary costs either way.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

struct Big {            // a kernel argument block of the engine's size (csf_dev.h: Dev, 1.3 KB): does its size cost at the boundary?
    float pad[320];
};

template <bool BIG>
__global__ __launch_bounds__(768) void busy_kernel(const float *in, float *out, int busy_ticks, Big big) {
    const unsigned long long t0 = wall_clock64();
    float v = in[(blockIdx.x * 64 + (threadIdx.x & 63)) & 16383];
    while ((long long)(wall_clock64() - t0) < busy_ticks) v = v * 1.0000001f + 1e-9f;     // (100 MHz: 100 ticks per microsecond)
    if (BIG) v += big.pad[threadIdx.x & 255] + big.pad[319];       // (read: first from its last line, as a struct's late members are)
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = v;
}

int main(int argc, char **argv) {
    const double busy_us = argc > 1 ? atof(argv[1]) : 7.0;
    const int busy = (int)(busy_us * 100.0), N = 4096, NODES = 64;
    float *a, *b;
    CHECK(hipMalloc(&a, 16384 * sizeof(float)));
    CHECK(hipMalloc(&b, 16384 * sizeof(float)));
    CHECK(hipMemset(a, 0, 16384 * sizeof(float)));
    CHECK(hipMemset(b, 0, 16384 * sizeof(float)));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const bool big_args = argc > 2 && atoi(argv[2]) != 0;
    Big big{};
    auto launch = [&](int i) {
        if (big_args) hipLaunchKernelGGL(busy_kernel<true>, dim3(256), dim3(768), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, busy, big);
        else hipLaunchKernelGGL(busy_kernel<false>, dim3(256), dim3(768), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, busy, big);
    };
    for (int i = 0; i < 256; i++) launch(i);
    CHECK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) launch(i);
    CHECK(hipStreamSynchronize(st));
    const double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < NODES; i++) launch(i);
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 4; r++) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipStreamSynchronize(st));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < N / NODES; r++) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipStreamSynchronize(st));
    const double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    printf("%s busy %.1f us per kernel: %d dependent launches on a stream %.2f us each (boundary %.2f); as graphs of %d nodes %.2f us each (boundary %.2f)\n",
           big_args ? "1.3 KB of arguments read," : "arguments unread,", busy_us, N, us_stream, us_stream - busy_us, NODES, us_graph, us_graph - busy_us);
    return 0;
}
