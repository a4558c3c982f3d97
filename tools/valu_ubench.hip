// VALU issue-rate microbenchmark for gfx950: wave-instructions per SIMD-cycle for the ops the pair kernel uses.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_ubench valu_ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 2048
typedef float float2v __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(float *out, float a, float b) {
    float x[8];
    float2v p[8];
    unsigned long long m = 0x5555aaaa3333ccccull, m2 = 0;
    float2v pa = {a, b};
    m = __builtin_amdgcn_readfirstlane((int)m) | ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32);
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = a + i + threadIdx.x * 1e-3f; p[i] = {x[i], x[i] + 1.f}; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i+1)&7]), "v"(p[(i+2)&7]));
            if (OP == 2) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i]));
            if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            if (OP == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a));
            if (OP == 6) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));
            if (OP == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
            if (OP == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i+1)&7]));
            if (OP == 9) asm volatile("v_cmp_ge_f32 vcc, %0, %1" :: "v"(x[i]), "v"(a) : "vcc");
            if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(b));
            if (OP == 11) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (OP == 12) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(m));
            if (OP == 13) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (OP == 14) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (OP == 15) asm volatile("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m2) : "v"(x[i]), "v"(a));
            if (OP == 16) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(x[i]) : "s"((unsigned)m));
            if (OP == 17) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x[i]) : "s"(a));
            if (OP == 18) asm volatile("v_fma_f32 %0, %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (OP == 19) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (OP == 20) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(x[i]) : "s"(a));
            if (OP == 21) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i+1)&7]));
            if (OP == 22) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : );
            if (OP == 23) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
            if (OP == 24) asm volatile("v_fma_f32 %0, |%0|, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (OP == 25) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "s"(pa), "v"(p[(i+2)&7]));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(t1 - t0) * 0.f + (float)m2 * 0.f;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((unsigned long long *)out)[1 << 20] = t1 - t0;
}

template <int OP>
void run(const char *name, float *d, int waves_per_simd) {
    int cus = 256;
    dim3 g(cus * waves_per_simd), b(256);  // 4 waves per block -> one per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<g, b>>>(d, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<g, b>>>(d, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc; hipMemcpy(&cyc, ((unsigned long long *)d) + (1 << 20), 8, hipMemcpyDeviceToHost);
    double instr_per_simd = (double)ITER * 8 * waves_per_simd;  // wave-instructions issued on one SIMD
    printf("%-14s waves/SIMD=%d  time=%8.1f us  -> %5.2f ns per wave-instr per SIMD = %5.2f cycles @2.4GHz ; s_memtime-cycles/instr(one wave)=%.2f\n",
           name, waves_per_simd, ms * 1e3, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4, (double)cyc / (ITER * 8));
}

int main() {
    float *d; hipMalloc(&d, (1 << 23) + 64);
    for (int w : {2, 8}) {
        run<0>("v_fma_f32", d, w);
        run<10>("v_fma_f32(sgpr)", d, w);
        run<1>("v_pk_fma_f32", d, w);
        run<4>("v_mul_f32", d, w);
        run<8>("v_pk_mul_f32", d, w);
        run<2>("v_rsq_f32", d, w);
        run<6>("v_sqrt_f32", d, w);
        run<7>("v_rcp_f32", d, w);
        run<3>("v_exp_f32", d, w);
        run<5>("v_cndmask_b32", d, w);
        run<9>("v_cmp_ge_f32", d, w);
        run<11>("v_med3_f32", d, w);
        run<12>("cndmask_e64_sgpr", d, w);
        run<22>("cndmask_e32_vcc", d, w);
        run<13>("v_add_f32_e32", d, w);
        run<14>("v_fmac_f32_e32", d, w);
        run<15>("v_cmp_e64->sgpr", d, w);
        run<16>("v_mbcnt_lo", d, w);
        run<17>("v_mul_e32 sgpr", d, w);
        run<18>("v_fma 2reads", d, w);
        run<19>("v_max_f32_e32", d, w);
        run<20>("v_sub_e32 sgpr", d, w);
        run<21>("v_pk_add_f32", d, w);
        run<23>("v_mov_dpp", d, w);
        run<24>("v_fma |abs|", d, w);
        run<25>("v_pk_fma sgpr", d, w);
        printf("\n");
    }
    return 0;
}
