// How much does workgroup turnover cost on MI355X?  The same total fp32-MFMA work is launched as many short workgroups
// (one "tile" each, like the convolution: 64x128 tile, K = 576 -> 576 MFMAs per wave) or as fewer long ones.
//   hipcc --offload-arch=gfx950 -O3 tools/exp_dispatch.hip -o tools/bin/exp_dispatch && tools/bin/exp_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int LDSKB>
__global__ __launch_bounds__(256) void mfma_only(float* out, int n_mfma, int tiles_per_wg, int epilogue) {
    __shared__ float lds[LDSKB * 256 + 256];
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const float a = 1.0f + threadIdx.x, b = 2.0f + threadIdx.x;
    lds[threadIdx.x] = a; if (a == -1.0f) out[0] = lds[(threadIdx.x * 7) % (LDSKB * 256 + 256)];
    for (int t = 0; t < tiles_per_wg; ++t) {
        for (int i = 0; i < n_mfma; i += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        }
        if (epilogue) {     // a tile's worth of stores: 32 coalesced dword stores per lane
            float* o = out + ((size_t)(blockIdx.x * tiles_per_wg + t) * 256 + threadIdx.x) * 32;
            for (int r = 0; r < 16; ++r) { o[r] = acc0[r]; o[16 + r] = acc1[r]; }
        }
    }
    if (!epilogue && acc0[0] + acc1[0] == 12345.678f) out[threadIdx.x] = acc0[0];
}

template <int LDSKB> void run(const char* name, int total_tiles, int n_mfma, int tiles_per_wg, int epilogue, float* d) {
    const int grid = total_tiles / tiles_per_wg;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) mfma_only<LDSKB><<<grid, 256>>>(d, n_mfma, tiles_per_wg, epilogue);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int it = 20;
    for (int i = 0; i < it; ++i) mfma_only<LDSKB><<<grid, 256>>>(d, n_mfma, tiles_per_wg, epilogue);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
    const double flop = (double)total_tiles * 4 /*waves*/ * n_mfma * 4096.0;
    printf("%-28s grid %5d x %2d tiles x %4d MFMA/wave, LDS %2d KB, stores %d: %8.1f us  %6.1f TF\n", name, grid, tiles_per_wg, n_mfma, LDSKB, epilogue, ms * 1e3, flop / ms / 1e9);
}

int main() {
    float* d; CK(hipMalloc(&d, (size_t)1280 * 256 * 32 * 4 * 2));
    run<0>("tile per WG", 1280, 576, 1, 0, d);
    run<0>("tile per WG + stores", 1280, 576, 1, 1, d);
    run<32>("tile per WG, 32 KB LDS", 1280, 576, 1, 0, d);
    run<64>("tile per WG, 64 KB LDS", 1280, 576, 1, 0, d);
    run<0>("5 tiles per WG (256 WGs)", 1280, 576, 5, 0, d);
    run<0>("5 tiles per WG + stores", 1280, 576, 5, 1, d);
    run<32>("5 tiles/WG 32KB + stores", 1280, 576, 5, 1, d);
    run<0>("2.5 tiles (512 WGs)", 1280, 1440, 1, 0, d);
    run<0>("long: 256 WGs x 28800", 256, 28800, 1, 0, d);
    run<0>("long: 512 WGs x 14400", 512, 14400, 1, 0, d);
    run<0>("1 tile, K=1152 (64x64)", 1280, 576, 1, 0, d);
    return 0;
}
