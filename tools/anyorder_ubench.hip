// Does hipExtAnyOrderLaunch let a kernel start beside its predecessor in the SAME stream on gfx950?
// Two kernels of one wave per CU that each spin ~50 us and stamp wall_clock64 at start and end.
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_ubench.hip -o build/anyorder_ubench && build/anyorder_ubench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>

__global__ void spin(uint64_t *stamps, int which, uint64_t ticks) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        atomicMin((unsigned long long *)&stamps[2 * which], (unsigned long long)t0);
        atomicMax((unsigned long long *)&stamps[2 * which + 1], (unsigned long long)wall_clock64());
    }
}

int main() {
    uint64_t *st;
    hipMalloc(&st, 64);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rep = 0; rep < 4; rep++) {
        for (int mode = 0; mode < 3; mode++) {
            uint64_t init[4] = {~0ull, 0, ~0ull, 0};
            hipMemcpy(st, init, sizeof init, hipMemcpyHostToDevice);
            const uint64_t ticks = 5000;   // 100 MHz clock: 50 us
            hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, nullptr, nullptr, 0, st, 0, ticks);
            hipExtLaunchKernelGGL(spin, dim3(mode == 2 ? 2048 : 256), dim3(64), 0, s, nullptr, nullptr, mode >= 1 ? hipExtAnyOrderLaunch : 0, st, 1, ticks);
            hipStreamSynchronize(s);
            uint64_t h[4];
            hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost);
            printf("mode %d (%s): A %.1f us, B starts %.1f us after A starts, both done after %.1f us\n", mode,
                   mode ? "second launch with hipExtAnyOrderLaunch" : "plain", (h[1] - h[0]) / 100.0, ((double)h[2] - (double)h[0]) / 100.0,
                   (h[3] - h[0]) / 100.0);
        }
    }
    return 0;
}
