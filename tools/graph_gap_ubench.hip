// What does a hipGraph save between DEPENDENT launches on this GPU, and what do a large argument block, scratch memory and LDS cost
// at the boundary?  (measurement aid)   graph_gap [busy_us] [mode]
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

// MODE bits: 1 the argument block is read, 2 / 8 a scratch array (208 bytes per lane) touched by every lane / one lane per workgroup,
// 4 / 16 60 KB of LDS touched by every lane (with a barrier) / by one lane
template <int MODE>
__global__ __launch_bounds__(768) void busy_kernel(const float *in, float *out, int busy_ticks, Big big) {
    constexpr bool BIG = (MODE & 1) != 0;
    __shared__ float lds[(MODE & (4 | 16)) ? 15000 : 1];
    float stack[48];

    if (MODE & 4) lds[threadIdx.x] = in[threadIdx.x];
    if ((MODE & 16) && threadIdx.x == 0) lds[blockIdx.x & 1023] = in[0];      // (60 KB allocated, one word touched)
    const unsigned long long t0 = wall_clock64();
    float v = in[(blockIdx.x * 64 + (threadIdx.x & 63)) & 16383];
    // (scratch memory, 208 bytes per lane allocated: MODE 2 - every lane touches two of its words; MODE 8 - one lane per workgroup does)
    if ((MODE & 2) || ((MODE & 8) && threadIdx.x == 0)) {
        stack[threadIdx.x % 48] = v;
        stack[(threadIdx.x + 7) % 48] = 1.0f;
    }
    if ((MODE & 2) || ((MODE & 8) && threadIdx.x == 0)) v += stack[(int)(v * 1e-9f + (float)(threadIdx.x % 48)) % 48] + stack[(threadIdx.x + 7) % 48];   // (read back inside the busy time)
    while ((long long)(wall_clock64() - t0) < busy_ticks) v = v * 1.0000001f + 1e-9f;     // (100 MHz: 100 ticks per microsecond)
    if (MODE & 4) { __syncthreads(); v += lds[(threadIdx.x * 7) % 15000]; }
    if ((MODE & 16) && threadIdx.x == 0) v += lds[blockIdx.x & 1023];
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
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const bool big_args = (mode & 1) != 0;
    Big big{};
    auto launch = [&](int i) {
        const float *src = (i & 1) ? b : a;
        float *dst = (i & 1) ? a : b;
        switch (mode) {
        case 1: hipLaunchKernelGGL(busy_kernel<1>, dim3(256), dim3(768), 0, st, src, dst, busy, big); break;
        case 2: hipLaunchKernelGGL(busy_kernel<2>, dim3(256), dim3(768), 0, st, src, dst, busy, big); break;
        case 4: hipLaunchKernelGGL(busy_kernel<4>, dim3(256), dim3(768), 0, st, src, dst, busy, big); break;
        case 16: hipLaunchKernelGGL(busy_kernel<16>, dim3(256), dim3(768), 0, st, src, dst, busy, big); break;
        case 8: hipLaunchKernelGGL(busy_kernel<8>, dim3(256), dim3(768), 0, st, src, dst, busy, big); break;
        case 7: hipLaunchKernelGGL(busy_kernel<7>, dim3(256), dim3(768), 0, st, src, dst, busy, big); break;
        default: hipLaunchKernelGGL(busy_kernel<0>, dim3(256), dim3(768), 0, st, src, dst, busy, big); break;
        }
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
    printf("mode %d (%s) busy %.1f us per kernel: %d dependent launches on a stream %.2f us each (boundary %.2f); as graphs of %d nodes %.2f us each (boundary %.2f)\n",
           mode, big_args ? "1.3 KB of arguments read" : "arguments unread", busy_us, N, us_stream, us_stream - busy_us, NODES, us_graph, us_graph - busy_us);
    return 0;
}
