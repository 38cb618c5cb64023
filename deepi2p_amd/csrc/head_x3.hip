// The fused per-point head (per_point_pn of models/networks_united.py:57-74, applied at :188-197; coarse variant 736 -> 128 -> 128 -> P) on the
// bf16 matrix instructions with EXACT three-way fp32 operand splits ("bf16x3", gemm.hip / conv_x3.hip), WAVE-AUTONOMOUS: a wave owns all
// 128 channels of 32 points through
//   layer 0   y0 = relu(scale0 * (W0 x + sum_j w_j G[idx_j]) + shift0)       x = the K0 dense channels of the point (read straight from memory
//             into B-fragment layout: lane (point n, half h) loads channels 16 s + 8 h + 0..7 of its point and splits them in registers);
//             the gathered per-node products (layer 0 of the reference contracts 640 interpolated channels: W sum_k w_k f_k =
//             sum_k w_k (W f_k), contracted once per NODE by two small GEMMs) INITIALISE the accumulators -- from the frame's two node tables
//             resident in LDS (TAB_LDS: 2 x 128 nodes x 128 channels, rows padded against bank conflicts) or from memory;
//   layer 1   y1 = relu(scale1 * W1 y0 + shift1): its B operand is layer 0's output after one v_permlane32_swap per register pair
//             (rows 16 s + 8 h + 0..7 of a column live half in lane n, half in lane n + 32) and the same split;
//   layer 2   the P <= 4 outputs as fp32 dot products (each lane holds 64 of a column's 128 rows; the two halves meet through a lane swap).
// No activation ever leaves the registers, nothing is staged through LDS, there is no barrier after the tables are in place.  The weights are
// split ONCE per checkpoint into fragment order ([K-step][row tile][plane][lane] x 16 B: a wave's fragment load is one contiguous KB) and read
// from L1 / L2 a K-step ahead.
// Against the fp32-MFMA head (gemm.hip point_head_kernel: 128 x 64 LDS tile, 9.2 M v_mfma_f32_32x32x2_f32 per step, six 512-byte table rows
// per point through L2) this issues 6/16 of the matrix-pipe time and, with the tables in LDS, none of the 2 GB of gather traffic.
#include <stdint.h>

#include <type_traits>

#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HX_M = 128, HX_TM = 4, HX_TROW = HX_M + 4;      // table row stride in LDS (floats): 528 B, 16-byte aligned, off the 512-byte period

struct HxArgs {
    const float* src[2]; long long src_bs[2]; int src_rs[2]; int c0, K0;       // two dense sources f32[B, ch, N]; channels [0, c0) from src 0
    const u32x4_t* W0p; const u32x4_t* W1p;                                     // fragment-ordered splits of W0t[K0][128], W1t[128][128]
    const float* ss;                                                            // [4][128]: scale0, shift0, scale1, shift1
    const float* tab[2]; const int* idx[2]; const float* gw[2]; int nodes[2];   // node tables f32[B, nodes, 128], idx i32[B, N, 3], w f32[B, N, 3] | null
    const float* W2t; const float* sc2; const float* sh2;                       // [128][P], [P] | null
    float* out;                                                                 // f32[B, P, N]
    int relu0, relu1, relu2, P, N, nblk, total, parts;                          // nblk = 32-point blocks per frame; total = B * nblk; TAB_LDS: workgroups per frame
};

__device__ __forceinline__ float hx_hi16(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u); }
__device__ __forceinline__ unsigned hx_pack_hi(float x0, float x1) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
}
__device__ __forceinline__ void hx_split8(const float (&f)[8], u32x4_t (&p)[3]) {
    float r[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { r[i] = f[i] - hx_hi16(f[i]); q[i] = r[i] - hx_hi16(r[i]); }
    p[0] = u32x4_t{hx_pack_hi(f[0], f[1]), hx_pack_hi(f[2], f[3]), hx_pack_hi(f[4], f[5]), hx_pack_hi(f[6], f[7])};
    p[1] = u32x4_t{hx_pack_hi(r[0], r[1]), hx_pack_hi(r[2], r[3]), hx_pack_hi(r[4], r[5]), hx_pack_hi(r[6], r[7])};
    p[2] = u32x4_t{hx_pack_hi(q[0], q[1]), hx_pack_hi(q[2], q[3]), hx_pack_hi(q[4], q[5]), hx_pack_hi(q[6], q[7])};
}
__device__ __forceinline__ f32x16 hx_mma(const u32x4_t& a, const u32x4_t& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// TAB_LDS: the frame's node tables in LDS (one 8-wave workgroup per compute unit, two waves per SIMD); KS0 = K0 / 16 K-steps of layer 0; NW waves
// SYNC (TAB_LDS only): the workgroup's waves take every K-step together (one s_barrier each), so that the 12 KB weight panel of a K-step is
// fetched from L2 once per workgroup and served to the other seven waves by the L1
// DEEP (one wave per SIMD, 512 registers): a K-step's weight fragments are requested a whole K-step ahead (else plane by plane, as registers free up)
// LEAN (two waves per SIMD, 256 registers): the gathered rows are requested one row group ahead instead of two (48 registers less in epilogue 0)
// and the output layer's four sums are pinned row group by row group -- together: no scratch at 256 registers
template <bool TAB_LDS, int KS0, int NW, int MINW, bool SYNC, bool DEEP, bool LEAN = false>
__global__ __launch_bounds__(NW * 64, MINW) void point_head_x3_kernel(const HxArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ssl = lds;                                  // [4][128] scale / shift rows
    float* w2l = ssl + 4 * HX_M;                       // [128][4]  output layer, row stride 4
    float* tabl = w2l + 4 * HX_M;                      // TAB_LDS: [2][nodes][HX_TROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, h = lane >> 5;
    for (int i = tid; i < 4 * HX_M; i += NW * 64) {
        ssl[i] = a.ss[i];
        w2l[i] = (i & 3) < a.P ? a.W2t[(i >> 2) * a.P + (i & 3)] : 0.0f;
    }
    int g_first, g_stride, g_end, fb_fixed = 0;
    if (TAB_LDS) {
        // the workgroup belongs to ONE frame: its node tables come to LDS once, its waves walk the frame's blocks part by part
        const int fb = blockIdx.x / a.parts, part = blockIdx.x - fb * a.parts;
        fb_fixed = fb;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float* T = a.tab[t] + (long long)fb * a.nodes[t] * HX_M;
            float* L = tabl + (t == 0 ? 0 : a.nodes[0] * HX_TROW);
            for (int i = tid; i < a.nodes[t] * (HX_M / 4); i += NW * 64) {
                const int node = i / (HX_M / 4), c4 = i - node * (HX_M / 4);
                *reinterpret_cast<float4*>(L + node * HX_TROW + 4 * c4) = *reinterpret_cast<const float4*>(T + node * HX_M + 4 * c4);
            }
        }
        g_first = fb * a.nblk + part * NW + wave; g_stride = a.parts * NW; g_end = (fb + 1) * a.nblk;
    } else {
        g_first = blockIdx.x * NW + wave; g_stride = gridDim.x * NW; g_end = a.total;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t w0r = __builtin_amdgcn_make_buffer_rsrc((void*)a.W0p, 0, (a.K0 / 16) * HX_TM * 3 * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc((void*)a.W1p, 0, (HX_M / 16) * HX_TM * 3 * 1024, 0x00020000);

    int gi_n[6];                 // DEEP: the next block's inputs, in flight across the loop's back edge
    float gw_n[6], x_n[KS0][8];
    // SYNC: every wave runs the same number of trips (a wave past the frame's last block recomputes it and stores nothing)
    const int trips = SYNC ? (g_end - (g_first - wave) + g_stride - 1) / g_stride : 0;
    for (int it = 0, g0 = g_first; SYNC ? it < trips : g0 < g_end; ++it, g0 += g_stride) {
        const bool live = g0 < g_end;
        const int g = SYNC ? min(g0, g_end - 1) : g0;
        // (the scale / shift rows and the output layer are RE-READ from LDS in every block: hoisted out of this loop -- they are loop
        //  invariant -- they would occupy some 200 registers, i.e. scratch)
        asm volatile("" ::: "memory");
        int hq = 4 * h;                          // laundered per block: the 48 LDS row addresses derived from it are otherwise hoisted out of the loop -- and spilled
        asm volatile("" : "+v"(hq));
        const int fb = TAB_LDS ? fb_fixed : __builtin_amdgcn_readfirstlane(g / a.nblk);
        const int n = (g - fb * a.nblk) * 32 + nl, nc = min(n, a.N - 1);
        // ---- a block's inputs: the point's six neighbours (3 + 3: node index, interpolation weight) and layer 0's operands -- channels
        // 16 s + 8 h + e of the point, straight from memory (rows of src 0 up to c0, then src 1)
        auto request_nb = [&](int gg, int (&gi_)[6], float (&gw_)[6]) __attribute__((always_inline)) {
            const int fq = TAB_LDS ? fb_fixed : __builtin_amdgcn_readfirstlane(gg / a.nblk);
            const int nq = min((gg - fq * a.nblk) * 32 + nl, a.N - 1);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const long long o = ((long long)fq * a.N + nq) * 3 + j;
                    gi_[3 * t + j] = a.idx[t][o];
                    gw_[3 * t + j] = a.gw[t] ? a.gw[t][o] : 1.0f;
                }
        };
        auto x_load = [&](int gg, int s, float (&f)[8]) __attribute__((always_inline)) {
            const int fq = TAB_LDS ? fb_fixed : __builtin_amdgcn_readfirstlane(gg / a.nblk);
            const int nq = min((gg - fq * a.nblk) * 32 + nl, a.N - 1);
            const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src[0] + (long long)fq * a.src_bs[0]), 0, a.c0 * a.src_rs[0] * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src[1] + (long long)fq * a.src_bs[1]), 0, (a.K0 - a.c0) * a.src_rs[1] * 4, 0x00020000);
            const int v0 = (8 * h * a.src_rs[0] + nq) * 4, v1 = (8 * h * a.src_rs[1] + nq) * 4;
            const int c = 16 * s;
            if (c < a.c0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r0, v0, (c + e) * a.src_rs[0] * 4, 0));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, v1, (c - a.c0 + e) * a.src_rs[1] * 4, 0));
            }
        };
        int gi[6];
        float gwt[6];
        float xall[KS0][8];
        if (it == 0) {
            request_nb(g, gi_n, gw_n);
#pragma unroll
            for (int s = 0; s < KS0; ++s) x_load(g, s, x_n[s]);
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) { gi[s] = gi_n[s]; gwt[s] = gw_n[s]; }
#pragma unroll
        for (int s = 0; s < KS0; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) xall[s][e] = x_n[s][e];
        f32x16 acc[HX_TM];
#pragma unroll
        for (int i = 0; i < HX_TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        // One K-step = six products per row tile, grouped by the plane of the WEIGHT fragment (third, second, first plane: the small terms still
        // come first) so that a plane's registers are free after its last product: the next K-step's fragments of that plane are requested
        // right there (one register set and a bit; two sets -- 96 registers -- spilled at two waves per SIMD).  The partner wave of the SIMD
        // covers what latency is left.
        u32x4_t af[2][HX_TM][3];
        auto a_load = [&](int slot, const __amdgpu_buffer_rsrc_t& wr, int ks, int p) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < HX_TM; ++i) af[slot][i][p] = __builtin_amdgcn_raw_buffer_load_b128(wr, lane * 16, ((ks * HX_TM + i) * 3 + p) * 1024, 0);
        };
        // ks_next < 0: nothing to request
        auto kstep = [&](int slot, const u32x4_t (&bf)[3], const __amdgpu_buffer_rsrc_t& wr, int ks_next) __attribute__((always_inline)) {
#define DI2P_HX_PROD(QA, QB) _Pragma("unroll") for (int i = 0; i < HX_TM; ++i) acc[i] = hx_mma(af[slot][i][QA], bf[QB], acc[i]);
            if constexpr (SYNC) __builtin_amdgcn_s_barrier();
            if constexpr (DEEP) {
                if (ks_next >= 0) { a_load(slot ^ 1, wr, ks_next, 2); a_load(slot ^ 1, wr, ks_next, 1); a_load(slot ^ 1, wr, ks_next, 0); }
            }
            DI2P_HX_PROD(2, 0)
            if (!DEEP && ks_next >= 0) a_load(slot ^ 1, wr, ks_next, 2);
            DI2P_HX_PROD(1, 1) DI2P_HX_PROD(1, 0)
            if (!DEEP && ks_next >= 0) a_load(slot ^ 1, wr, ks_next, 1);
            DI2P_HX_PROD(0, 2) DI2P_HX_PROD(0, 1) DI2P_HX_PROD(0, 0)
            if (!DEEP && ks_next >= 0) a_load(slot ^ 1, wr, ks_next, 0);
#undef DI2P_HX_PROD
        };
        // ---- layer 0 (its operands were requested at the end of the previous block: only weight fragments are in flight here)
        {
#pragma unroll
            for (int p = 0; p < 3; ++p) a_load(0, w0r, 0, p);
            // the split of K-step s + 1 (about fifty vector instructions) rides between the matrix instructions of K-step s: two fragment sets,
            // one scheduling region per K-step laid out as "one matrix instruction, two vector instructions"
            u32x4_t bfr[2][3];
            hx_split8(xall[0], bfr[0]);
#pragma unroll
            for (int s = 0; s < KS0; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < KS0) hx_split8(xall[s + 1], bfr[(s + 1) & 1]);
                kstep(s & 1, bfr[s & 1], w0r, s + 1 < KS0 ? s + 1 : -1);
#pragma unroll
                for (int u = 0; u < 6 * HX_TM; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue 0 -> layer 1's operands (K-step s = 2 i + q: rows 32 i + 16 q + 8 h + 0..7 of the lane's column).  The gathered node products
        // t = sum_s w_s G_s[rows] (tables in order, neighbours in order: an fma chain from zero, as di2p_point_head forms it) join here, row
        // group by row group (4 rows x 6 neighbours = six 16-byte reads), group k + 1 requested before group k is reduced and no further ahead
        // (left alone, hipcc requests all 96 rows at once: 384 registers)
#pragma unroll
        for (int p = 0; p < 3; ++p) a_load(0, w1r, 0, p);
        const float* rp[6];
        {
            const float* T0 = TAB_LDS ? tabl : a.tab[0] + (long long)fb * a.nodes[0] * HX_M;
            const float* T1 = TAB_LDS ? tabl + a.nodes[0] * HX_TROW : a.tab[1] + (long long)fb * a.nodes[1] * HX_M;
            const int rs = TAB_LDS ? HX_TROW : HX_M;
#pragma unroll
            for (int s = 0; s < 6; ++s) rp[s] = (s < 3 ? T0 : T1) + gi[s] * rs + hq;
        }
        float4 q[LEAN ? 2 : 4][6];
        auto g_issue = [&](int k, float4 (&qq)[6]) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 6; ++s) qq[s] = *reinterpret_cast<const float4*>(rp[s] + 32 * (k >> 2) + 8 * (k & 3));
        };
        g_issue(0, q[0]);
        if (!LEAN) g_issue(1, q[1]);
        float f1[2 * HX_TM][8];
#pragma unroll
        for (int i = 0; i < HX_TM; ++i) {
            float v[16];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int k = 4 * i + gq;
                if (LEAN) {
                    // one row group is reduced per scheduling region while the next one is in flight
                    if (k + 1 < 4 * HX_TM) g_issue(k + 1, q[(k + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                } else if ((k & 1) == 0) {
                    // two row groups are reduced per scheduling region while the next two are in flight
                    if (k + 2 < 4 * HX_TM) g_issue(k + 2, q[(k + 2) & 3]);
                    if (k + 3 < 4 * HX_TM) g_issue(k + 3, q[(k + 3) & 3]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float4 (&qq)[6] = q[LEAN ? (k & 1) : (k & 3)];
                float4 t4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    t4.x = fmaf(gwt[s], qq[s].x, t4.x); t4.y = fmaf(gwt[s], qq[s].y, t4.y);
                    t4.z = fmaf(gwt[s], qq[s].z, t4.z); t4.w = fmaf(gwt[s], qq[s].w, t4.w);
                }
                const float4 sc = *reinterpret_cast<const float4*>(ssl + 32 * i + 8 * gq + hq);
                const float4 sh = *reinterpret_cast<const float4*>(ssl + HX_M + 32 * i + 8 * gq + hq);
                v[4 * gq + 0] = (acc[i][4 * gq + 0] + t4.x) * sc.x + sh.x; v[4 * gq + 1] = (acc[i][4 * gq + 1] + t4.y) * sc.y + sh.y;
                v[4 * gq + 2] = (acc[i][4 * gq + 2] + t4.z) * sc.z + sh.z; v[4 * gq + 3] = (acc[i][4 * gq + 3] + t4.w) * sc.w + sh.w;
                if (LEAN || (k & 1)) __builtin_amdgcn_sched_barrier(0);
            }
            if (a.relu0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.0f);
            }
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // X = rows 8 (2 q2) + 4 h' + j, Y = rows 8 (2 q2 + 1) + 4 h' + j of the tile; swap: X' = {X.lo, Y.lo}, Y' = {X.hi, Y.hi}.
                    // (inline assembly: through __builtin_amdgcn_permlane32_swap hipcc dropped the SECOND result here and used the first for both --
                    //  measured: rows 8 g + 4..7 of layer 1's operand held rows 8 g + 0..3; the s_nop covers the VALU write just before)
                    float X = v[8 * q2 + j], Y = v[8 * q2 + 4 + j];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(X), "+v"(Y));
                    f1[2 * i + q2][j] = X;
                    f1[2 * i + q2][4 + j] = Y;
                }
        }
        // ---- layer 1
#pragma unroll
        for (int i = 0; i < HX_TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        {
            u32x4_t bfr[2][3];
            hx_split8(f1[0], bfr[0]);
#pragma unroll
            for (int s = 0; s < 2 * HX_TM; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < 2 * HX_TM) hx_split8(f1[s + 1], bfr[(s + 1) & 1]);
                kstep(s & 1, bfr[s & 1], w1r, s + 1 < 2 * HX_TM ? s + 1 : -1);
#pragma unroll
                for (int u = 0; u < 6 * HX_TM; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the NEXT block's inputs (dense channels, neighbour indices and weights: from HBM) are requested HERE, behind layer 1's last weight
        // fragments: memory returns a wave's loads in order, so a request that sits in front of a K-step's fragments makes the wait for them a wait
        // for HBM (measured: with the requests issued a layer earlier nothing got faster).  They fly through epilogue 1 and across the back edge.
        __builtin_amdgcn_sched_barrier(0);
        {
            const int gn = min(g + g_stride, g_end - 1);
            request_nb(gn, gi_n, gw_n);
#pragma unroll
            for (int s = 0; s < KS0; ++s) x_load(gn, s, x_n[s]);
        }
        // ---- epilogue 1 and the output layer: each lane holds 64 of its column's 128 rows (fenced row group by row group: the 96 LDS reads of
        // this part, hoisted into layer 1's matrix instructions, were spilled)
        __builtin_amdgcn_sched_barrier(0);
        float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < HX_TM; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                if (gq & 1) __builtin_amdgcn_sched_barrier(0);
                const int row = 32 * i + 8 * gq + hq;
                const float4 sc = *reinterpret_cast<const float4*>(ssl + 2 * HX_M + row);
                const float4 sh = *reinterpret_cast<const float4*>(ssl + 3 * HX_M + row);
                float y[4] = {acc[i][4 * gq + 0] * sc.x + sh.x, acc[i][4 * gq + 1] * sc.y + sh.y, acc[i][4 * gq + 2] * sc.z + sh.z, acc[i][4 * gq + 3] * sc.w + sh.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (a.relu1) y[j] = fmaxf(y[j], 0.0f);
                    const float4 w = *reinterpret_cast<const float4*>(w2l + 4 * (row + j));
                    part[0] = fmaf(w.x, y[j], part[0]); part[1] = fmaf(w.y, y[j], part[1]);
                    part[2] = fmaf(w.z, y[j], part[2]); part[3] = fmaf(w.w, y[j], part[3]);
                }
                // (the four sums are pinned row group by row group: left alone at 256 registers, hipcc forms all 64 activations first and then
                //  the four dot products one after the other -- 140 bytes of scratch per lane, reloaded four times)
                if (LEAN) asm volatile("" : "+v"(part[0]), "+v"(part[1]), "+v"(part[2]), "+v"(part[3]));
            }
        // the two halves of a column meet through the LDS crossbar (ds_bpermute with lane ^ 32; hipcc folds the SUM of the two results of one
        // v_permlane32_swap into twice the first -- measured: outputs came out as 2 x the low half-wave's partial sum)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float other = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) * 4, __builtin_bit_cast(int, part[p])));
            float o = (hq ? other : part[p]) + (hq ? part[p] : other);          // low half's partial sum first, in both half-waves
            if (p < a.P) {
                if (a.sc2) o *= a.sc2[p];
                if (a.sh2) o += a.sh2[p];
                if (a.relu2) o = fmaxf(o, 0.0f);
                if (n < a.N && hq == 0 && live) a.out[((long long)fb * a.P + p) * a.N + n] = o;
            }
        }
    }
}

// Wt f32[K][128] (k-major) -> fragment order [K / 16][4 row tiles][3 planes][64 lanes] x 8 bf16: lane (row i = lane & 31, half = lane >> 5) of
// row tile t holds k = 16 s + 8 half + 0..7 of row 32 t + i.  One thread per (K-step, tile, lane).
__global__ __launch_bounds__(256) void head_x3_pack_kernel(const float* __restrict__ Wt, unsigned short* __restrict__ Wp, int K) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= (K / 16) * HX_TM * 64) return;
    const int lane = t & 63, tile = (t >> 6) & 3, s = t >> 8;
    const int row = 32 * tile + (lane & 31), k0 = 16 * s + 8 * (lane >> 5);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = Wt[(long long)(k0 + e) * HX_M + row];
        const float a1 = hx_hi16(v), r1 = v - a1, a2 = hx_hi16(r1), r2 = r1 - a2;
        const float pl[3] = {a1, a2, r2};
#pragma unroll
        for (int p = 0; p < 3; ++p)
            Wp[((((long long)s * HX_TM + tile) * 3 + p) * 64 + lane) * 8 + e] = (unsigned short)(__builtin_bit_cast(unsigned, pl[p]) >> 16);
    }
}

}  // namespace

extern "C" long long di2p_head_x3_packed_bytes(int K) { return K >= 16 && K % 16 == 0 ? (long long)(K / 16) * HX_TM * 3 * 1024 : 0; }

// Wt f32[K][128] (the [K, M] layout of every pointwise layer; K % 16 == 0) -> the fragment-ordered split operand of di2p_point_head_x3.
extern "C" int di2p_head_x3_pack(const float* Wt, int K, void* Wp, void* stream) {
    DI2P_CHECK_ARG(Wt && Wp && K >= 16 && K % 16 == 0, "needs K % 16 == 0 (and 128 output channels)");
    DI2P_CHECK_ARG(((uintptr_t)Wp & 15) == 0, "packed weights must be 16-byte aligned");
    hipLaunchKernelGGL(head_x3_pack_kernel, dim3(di2p_cdiv((long long)(K / 16) * HX_TM * 64, 256)), dim3(256), 0, (hipStream_t)stream, Wt,
                       (unsigned short*)Wp, K);
    DI2P_RETURN_LAUNCH();
}

// The coarse per-point head in one launch on the bf16 matrix instructions (exact three-way splits, fp32 accumulation):
//   out f32[B,P,N] = L2( relu?(scale1 * W1 relu?(scale0 * (W0 cat(src0, src1) + sum_j w_j G_a[idx_a_j] + sum_j w_j G_b[idx_b_j]) + shift0) + shift1) )
// h: see di2p_head_x3_t (include/deepi2p_hip.h).  Hidden width 128, P <= 4, two dense sources whose channel counts are multiples of 16,
// three neighbours per table, at most 128 + 128 nodes for the LDS-resident tables (else the tables are read from memory).
extern "C" int di2p_point_head_x3(const di2p_head_x3_t* hd, float* out, int B, int N, void* stream) {
    DI2P_CHECK_ARG(hd && out, "null pointer");
    DI2P_CHECK_ARG(B >= 0 && N >= 1, "bad size");
    DI2P_CHECK_ARG(hd->src[0] && hd->src[1] && hd->W0p && hd->W1p && hd->scale_shift && hd->W2t, "null operand");
    DI2P_CHECK_ARG(hd->channels[0] >= 16 && hd->channels[0] % 16 == 0 && hd->channels[1] >= 16 && hd->channels[1] % 16 == 0, "source channels must be multiples of 16");
    DI2P_CHECK_ARG(hd->P >= 1 && hd->P <= 4, "1..4 outputs");
    for (int t = 0; t < 2; ++t) DI2P_CHECK_ARG(hd->tab[t] && hd->idx[t] && hd->nodes[t] >= 1, "both gathered tables (3 neighbours each) are required");
    DI2P_CHECK_ARG(((uintptr_t)hd->W0p & 15) == 0 && ((uintptr_t)hd->W1p & 15) == 0 && ((uintptr_t)hd->tab[0] & 15) == 0 && ((uintptr_t)hd->tab[1] & 15) == 0 &&
                   ((uintptr_t)hd->scale_shift & 15) == 0, "16-byte alignment of the packed weights, the tables and the scale / shift rows");
    for (int t = 0; t < 2; ++t)
        DI2P_CHECK_ARG((long long)hd->channels[t] * hd->row_stride[t] * 4 < (1ll << 31), "per-frame source extent must fit 31 bits");
    if (B == 0) return 0;
    HxArgs a{};
    for (int t = 0; t < 2; ++t) {
        a.src[t] = hd->src[t]; a.src_bs[t] = hd->batch_stride[t]; a.src_rs[t] = hd->row_stride[t];
        a.tab[t] = hd->tab[t]; a.idx[t] = hd->idx[t]; a.gw[t] = hd->w[t]; a.nodes[t] = hd->nodes[t];
    }
    a.c0 = hd->channels[0]; a.K0 = hd->channels[0] + hd->channels[1];
    a.W0p = (const u32x4_t*)hd->W0p; a.W1p = (const u32x4_t*)hd->W1p; a.ss = hd->scale_shift;
    a.W2t = hd->W2t; a.sc2 = hd->scale2; a.sh2 = hd->shift2; a.out = out;
    a.relu0 = hd->relu0; a.relu1 = hd->relu1; a.relu2 = hd->relu2; a.P = hd->P; a.N = N;
    a.nblk = di2p_cdiv(N, 32); a.total = B * a.nblk;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds_small = (size_t)8 * HX_M * sizeof(float);
    const size_t lds_tab = lds_small + (size_t)(hd->nodes[0] + hd->nodes[1]) * HX_TROW * sizeof(float);
    const long long opt = di2p_opt(DI2P_OPT_HEAD_X3_TAB);      // 0: tables from memory, 1 (default) / 2: in LDS when they fit (eight / four waves)
    DI2P_CHECK_ARG(a.K0 == 96, "this build instantiates the head for 96 dense channels (32 + 64: first_pointnet + second_pointnet)");
    if (opt != 0 && lds_tab <= 160 * 1024) {
        // whole workgroups per frame: enough of them to fill the chip, each with at least two passes over its waves
        const int cus = di2p_cu_count();
        int parts = di2p_cdiv(cus, B);
        while (parts > 1 && a.nblk < parts * 16) --parts;
        a.parts = parts;
        // Eight waves per workgroup, two per SIMD (256 registers, no scratch in the LEAN form): one wave's epilogues run under the other's matrix
        // instructions -- 242 us alone against 339 for four waves with 512 registers (knob value 2; the eight-wave form with the deeper request
        // rings spilled 368 bytes per lane and took 385-430 us).  In the 8-stream pipeline the two are within 0.5 % of each other.
        if (opt == 2) {
            (void)hipFuncSetAttribute((const void*)point_head_x3_kernel<true, 6, 4, 1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tab);
            hipLaunchKernelGGL((point_head_x3_kernel<true, 6, 4, 1, false, true>), dim3(B * parts), dim3(256), lds_tab, st, a);
        } else {
            (void)hipFuncSetAttribute((const void*)point_head_x3_kernel<true, 6, 8, 2, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tab);
            hipLaunchKernelGGL((point_head_x3_kernel<true, 6, 8, 2, false, false, true>), dim3(B * parts), dim3(512), lds_tab, st, a);
        }
    } else {
        const int grid = (int)(a.total / 4 < 1ll * di2p_cu_count() ? di2p_cdiv(a.total, 4) : di2p_cu_count());
        hipLaunchKernelGGL((point_head_x3_kernel<false, 6, 4, 1, false, true>), dim3(grid), dim3(256), lds_small, st, a);
    }
    DI2P_RETURN_LAUNCH();
}
