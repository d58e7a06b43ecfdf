// Semantics of the DPP wave shifts on gfx950 (v_mov_b32_dpp wave_shr:1 / wave_shl:1 with bound_ctrl off: a lane without
// a source keeps `old`).  Build: hipcc --offload-arch=gfx950 -O2 tools/micro/wave_shift.hip -o tools/micro/wave_shift
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *o)
{
    const int lane = threadIdx.x;
    const float v = (float)lane;
    const int inf = 0x7f800000;
    o[lane] = __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(v), 0x138, 0xf, 0xf, false));        // wave_shr:1
    o[64 + lane] = __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(v), 0x130, 0xf, 0xf, false));   // wave_shl:1
}
int main()
{
    float *d, h[128];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 2;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 3;
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const float want_r = i == 0 ? __builtin_huge_valf() : (float)(i - 1), want_l = i == 63 ? __builtin_huge_valf() : (float)(i + 1);
        if (h[i] != want_r || h[64 + i] != want_l) { ++bad; printf("lane %d: shr %g (want %g) shl %g (want %g)\n", i, h[i], want_r, h[64 + i], want_l); }
    }
    printf("wave_shr:1 -> lane i reads lane i-1 (lane 0 keeps old), wave_shl:1 -> lane i reads lane i+1 (lane 63 keeps old): %s\n", bad ? "NO" : "yes");
    return bad ? 1 : 0;
}
