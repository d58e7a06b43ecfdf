// Throughput of fp64 division / sqrt forms on gfx950 (investigation aid; hipcc --offload-arch=gfx950 -O3 -ffp-contract=off)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP> __global__ void k(double *out, double seed, int iters)
{
    double a = seed + threadIdx.x * 1e-3, b = 1.5 + blockIdx.x * 1e-6, acc = 0.0;
    for (int i = 0; i < iters; ++i) {
        double r;
        if (OP == 0) r = __builtin_fma(a, b, acc);                       // baseline fma
        else if (OP == 1) r = a / b;                                      // IEEE division
        else if (OP == 2) r = __builtin_sqrt(a);                          // IEEE sqrt
        else if (OP == 3) { double x = __builtin_amdgcn_rcp(b); double e = __builtin_fma(-b, x, 1.0); x = __builtin_fma(x, e, x);
                            e = __builtin_fma(-b, x, 1.0); x = __builtin_fma(x, e, x); r = a * x; }          // rcp + 2 Newton
        else if (OP == 4) { double y = __builtin_amdgcn_rsq(a); double h = 0.5 * y, g = a * y; double e = __builtin_fma(-h, g, 0.5);
                            g = __builtin_fma(g, e, g); h = __builtin_fma(h, e, h); e = __builtin_fma(-h, g, 0.5); r = __builtin_fma(g, e, g); }  // rsq + 2 Newton
        else r = __builtin_amdgcn_rcp(b);                                 // bare v_rcp_f64
        acc += r; a += 1e-9; b += 1e-9;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int OP> void run(const char *name, double *d)
{
    const int blocks = 256 * 8, iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2.0, iters);
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2.0, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters;
    printf("%-18s %8.3f ms  %.2f Gop/s  (%.1f SIMD-cycles per wave-op incl. loop overhead)\n", name, ms, ops / ms * 1e-6,
           ms * 1e-3 * 2.4e9 * 1024 / (ops / 64));
}
int main()
{
    double *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(double));
    run<0>("fma", d); run<1>("a / b", d); run<2>("sqrt(a)", d); run<3>("rcp + 2 NR", d); run<4>("rsq + 2 NR", d); run<5>("bare rcp", d);
    hipFree(d); return 0;
}
