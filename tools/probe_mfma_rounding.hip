// How do the bf16 matrix instructions of gfx950 round?  One product a*b (bf16 operands) is added to an fp32 C whose ulp is 2^-23 (C = +-1.0):
//   a*b = f * 2^-23 for f in {0.25, 0.5, 0.75, 1.25, 1.5, 1.75} and both signs of C.   RNE -> C moves when f > 0.5 (ties to even);
//   truncation (toward zero) -> |result| never exceeds the exact |sum| ...
// Also: two products p1 = 1.0 (as 1 * 1) and p2 = f * 2^-23 in ONE instruction with C = 0 (is the internal sum of products exact before the
// single rounding?), for v_mfma_f32_32x32x16_bf16, v_mfma_f32_16x16x32_bf16 and, as the control, v_mfma_f32_32x32x2_f32.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_rounding.hip -o tools/bin/probe_mfma_rounding && tools/bin/probe_mfma_rounding
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __bf16 tobf(float x) { unsigned u = __builtin_bit_cast(unsigned, x) >> 16; unsigned short s = (unsigned short)u; return __builtin_bit_cast(__bf16, s); }

// out[0]: 32x32x16 bf16, out[1]: 16x16x32 bf16, out[2]: 32x32x2 f32; element (0,0) of D
__global__ void probe(float c0, float a0, float b0, float a1, float b1, float* out) {
    const int lane = threadIdx.x;
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = tobf(0.0f); b[i] = tobf(0.0f); }
    if (lane == 0) { a[0] = tobf(a0); b[0] = tobf(b0); a[1] = tobf(a1); b[1] = tobf(b1); }      // row 0 / col 0, k = 0 and k = 1
    f32x16 c16; for (int i = 0; i < 16; ++i) c16[i] = 0.0f;
    f32x4 c4; for (int i = 0; i < 4; ++i) c4[i] = 0.0f;
    if (lane == 0) { c16[0] = c0; c4[0] = c0; }
    f32x16 d16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16, 0, 0, 0);
    f32x4 d4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4, 0, 0, 0);
    // f32 control: k = 0 in lanes 0..31, k = 1 in lanes 32..63
    float fa = lane == 0 ? a0 : (lane == 32 ? a1 : 0.0f), fb = lane == 0 ? b0 : (lane == 32 ? b1 : 0.0f);
    f32x16 e16 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c16, 0, 0, 0);
    if (lane == 0) { out[0] = d16[0]; out[1] = d4[0]; out[2] = e16[0]; }
}

int main() {
    float* out; hipMalloc(&out, 12);
    const float ulp = ldexpf(1.0f, -23);
    const float fs[6] = {0.25f, 0.5f, 0.75f, 1.25f, 1.5f, 1.75f};
    printf("single product f*ulp added to C (results in ulps of 1.0 relative to C):   32x32x16_bf16  16x16x32_bf16  32x32x2_f32 | RNE expects\n");
    for (int sc = 0; sc < 2; ++sc)
        for (int sp = 0; sp < 2; ++sp)
            for (int i = 0; i < 6; ++i) {
                const float c = sc ? -1.0f : 1.0f, f = fs[i] * (sp ? -1.0f : 1.0f);
                // f * 2^-23 = (f) * 2^-23 with bf16-exact factors: f has <= 3 significant bits
                probe<<<1, 64>>>(c, f, ulp, 0.0f, 0.0f, out);
                float h[3]; hipMemcpy(h, out, 12, hipMemcpyDeviceToHost);
                const double exact = (double)c + (double)f * ulp;
                const float rne = (float)exact;
                // below |1.0| the ulp halves: report in units of 2^-24
                printf("C=%+.0f f=%+5.2f : %+8.2f %+8.2f %+8.2f | %+8.2f   (units of 2^-24 from C)\n", c, f, (h[0] - c) / (ulp / 2), (h[1] - c) / (ulp / 2), (h[2] - c) / (ulp / 2), (rne - c) / (ulp / 2));
            }
    printf("two products in one instruction, p1 = 1.0, p2 = f*ulp, C = 0:\n");
    for (int i = 0; i < 6; ++i)
        for (int sp = 0; sp < 2; ++sp) {
            const float f = fs[i] * (sp ? -1.0f : 1.0f);
            probe<<<1, 64>>>(0.0f, 1.0f, 1.0f, f, ulp, out);
            float h[3]; hipMemcpy(h, out, 12, hipMemcpyDeviceToHost);
            const float rne = (float)(1.0 + (double)f * ulp);
            printf("f=%+5.2f : %+8.2f %+8.2f %+8.2f | %+8.2f   (units of 2^-24 from 1.0)\n", f, (h[0] - 1.0f) / (ulp / 2), (h[1] - 1.0f) / (ulp / 2), (h[2] - 1.0f) / (ulp / 2), (rne - 1.0f) / (ulp / 2));
        }
    printf("three-term cancellation inside one instruction: C = 1.0, p1 = -1.0, p2 = f*ulp (exact answer f*ulp):\n");
    for (int i = 0; i < 6; ++i) {
        probe<<<1, 64>>>(1.0f, -1.0f, 1.0f, fs[i], ulp, out);
        float h[3]; hipMemcpy(h, out, 12, hipMemcpyDeviceToHost);
        printf("f=%+5.2f : %+10.4f %+10.4f %+10.4f   (units of 2^-23; exact = f)\n", fs[i], h[0] / ulp, h[1] / ulp, h[2] / ulp);
    }
    return 0;
}
