// The head of the image branch -- conv 7x7 / stride 2 / pad 3 (3 -> 64), folded BatchNorm, ReLU, max-pool 3x3 / stride 2 / pad 1
// (models/resnet.py:137-141,197-201) -- as ONE launch on the bf16 matrix instructions with the exact three-way fp32 split of conv_x3.hip /
// gemm.hip ("bf16x3": six bf16 products per fp32 product, fp32 accumulation).  The 64 x OH x OW activation between the convolution and the
// pool (168 MB per 32-frame step at 160 x 512: written by stem.hip, read back by the pool kernel) never leaves the compute unit.
//
//   K layout: k = ((ci, ky) pair, kx padded 7 -> 8 with a zero weight): 21 pairs; one K-step of v_mfma_f32_32x32x16_bf16 is two pairs (the lane's
//            k-group = lane / 32 picks the pair), 11 K-steps (the 22nd pair has zero weights).  A lane's eight consecutive k of a B fragment
//            are then eight CONSECUTIVE input columns 2 ox - 3 ... 2 ox + 4 of one input row and channel: with the input rows kept in LDS as
//            three bf16 planes, left-padded by 3, a B fragment is 16 bytes at byte offset 4 ox of a row -- no im2col, no parity split.
//   weights: split once per checkpoint into fragment order [K-step][channel tile of 32][plane][lane] x 16 bytes (di2p_stem_x3_pack) and read from
//            L2 as A fragments, two K-steps ahead (66 KB: every workgroup reads the same lines).
//   input:   a workgroup owns `prw` pooled rows of one frame over the FULL width (no horizontal halo: the pool's left neighbour of column 0 is
//            padding) and walks over its 2 prw + 1 convolution rows; convolution row cr needs input rows 2 cr - 3 ... 2 cr + 3 of the three
//            channels: a ring of nine input rows (seven live + the two the next row adds) in LDS, split while they are staged.  The two new
//            rows are requested at the end of the previous convolution row and written between the matrix instructions of this one.
//   a wave:  64 convolution columns x 64 channels = 2 x 2 tiles; 264 matrix instructions per convolution row.
//   pool:    relu(scale * acc + shift) of the row goes to an LDS tile [channel][column]; a thread owns one pooled column and 32 channels: the
//            maximum over columns 2c-1, 2c, 2c+1 from the tile, the maximum over the three rows in its registers (the last row of one pooled
//            row is the first of the next); pooled rows leave as 512-byte row pieces.  The tile of row cr is pooled BETWEEN the matrix
//            instructions of row cr + 1; two barriers per row (tile free / tile and ring complete).
#include <stdint.h>

#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SX_CO = 64, SX_PAIRS = 21, SX_KSTEPS = 11, SX_RING = 9, SX_ITEMS = 4;
constexpr int SX_WP_BYTES = SX_KSTEPS * 2 * 3 * 64 * 16;
constexpr int SX_OOB = 0x40000000;

__device__ __forceinline__ float sx_hi16(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u); }
__device__ __forceinline__ unsigned sx_pack_hi(float x0, float x1) {      // bf16(x0) in the low half, bf16(x1) in the high half (truncation)
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
}

struct SxArgs {
    const float* x; const u32x4_t* Wp; const float* scale; const float* shift; float* y;
    int H, W, OH, OW, PH, PWo;      // input, convolution output, pooled output
    int prw, tiles;                 // pooled rows per workgroup, workgroups per frame
    int LW, LDT;                    // LDS row length of an input row (bf16 elements, W + 8), of a tile row (floats, OW + 4)
};

// weight f32[64][3][7][7] -> fragment order: entry ((s*2 + mt)*3 + plane)*64 + lane, lane = (channel mt*32 + lane%32, pair 2s + lane/32), eight
// kx (the eighth: zero)
__global__ __launch_bounds__(256) void stem_x3_pack_kernel(const float* __restrict__ w, u32x4_t* __restrict__ Wp) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= SX_KSTEPS * 2 * 64) return;
    const int lane = t & 63, mt = (t >> 6) & 1, s = t >> 7;
    const int c = mt * 32 + (lane & 31), p = 2 * s + (lane >> 5);
    float f[8], r[8], q[8];
#pragma unroll
    for (int kx = 0; kx < 8; ++kx) {
        f[kx] = (p < SX_PAIRS && kx < 7) ? w[(c * SX_PAIRS + p) * 7 + kx] : 0.0f;
        r[kx] = f[kx] - sx_hi16(f[kx]);
        q[kx] = r[kx] - sx_hi16(r[kx]);
    }
    u32x4_t* d = Wp + (long long)(s * 2 + mt) * 3 * 64 + lane;
    d[0] = u32x4_t{sx_pack_hi(f[0], f[1]), sx_pack_hi(f[2], f[3]), sx_pack_hi(f[4], f[5]), sx_pack_hi(f[6], f[7])};
    d[64] = u32x4_t{sx_pack_hi(r[0], r[1]), sx_pack_hi(r[2], r[3]), sx_pack_hi(r[4], r[5]), sx_pack_hi(r[6], r[7])};
    d[128] = u32x4_t{sx_pack_hi(q[0], q[1]), sx_pack_hi(q[2], q[3]), sx_pack_hi(q[4], q[5]), sx_pack_hi(q[6], q[7])};
}

__global__ __launch_bounds__(256, 1) void stem_x3_kernel(const SxArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, cl = lane >> 5;
    const int b = blockIdx.x / a.tiles, tile = blockIdx.x - b * a.tiles;
    const int r0 = tile * a.prw, r1 = min(r0 + a.prw, a.PH);
    const int cr_first = max(2 * r0 - 1, 0), cr_last = 2 * r1 - 1;
    const int rowb = a.LW * 2;                                       // bytes of one (slot, channel, plane) row
    const int slotb = 9 * rowb;                                      // bytes of one ring slot (3 channels x 3 planes)
    unsigned char* ring = smem;
    float* T = reinterpret_cast<float*>(smem + SX_RING * slotb);
    const int dump = SX_RING * slotb + SX_CO * a.LDT * 4;            // 16 bytes behind the tile: where the surplus staging items land
    const bool mma_wave = wave * 64 < a.OW;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)b * 3 * a.H * a.W), 0, 3 * a.H * a.W * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.Wp, 0, SX_WP_BYTES, 0x00020000);

    // ---- staging items of this thread (two input rows x three channels x LW / 4 column quads; LDS column L = input column + 3)
    const int QPR = a.LW / 4;
    unsigned st_off[SX_ITEMS][4];            // (unsigned: column part + row part may both be the out-of-range constant)
    int st_l[SX_ITEMS], st_rr[SX_ITEMS];
    bool st_ok[SX_ITEMS];
#pragma unroll
    for (int it = 0; it < SX_ITEMS; ++it) {
        const int q = tid + 256 * it, rc = q / QPR, qq = q - rc * QPR, rr = rc / 3, ci = rc - 3 * rr;
        st_ok[it] = q < 6 * QPR;
        st_rr[it] = rr;
        st_l[it] = ci * 3 * rowb + qq * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {       // byte offset of input column 4 qq - 3 + e in row 0 of channel ci; out of range: the load returns 0
            const int icol = 4 * qq - 3 + e;
            st_off[it][e] = (st_ok[it] && icol >= 0 && icol < a.W) ? (unsigned)((ci * a.H * a.W + icol) * 4) : (unsigned)SX_OOB;
        }
    }
    float raw[SX_ITEMS][4];
    // (offsets are SUMS of a column part and a row part, either of which may be the out-of-range constant: no branches -- hipcc turned
    //  `ok ? offset : OOB` on a conjunction into a ladder of exec-mask branches around single loads)
    auto stage_load = [&](float (&raw)[SX_ITEMS][4], int irow_new) __attribute__((always_inline)) {
        const unsigned radd0 = (irow_new >= 0 && irow_new < a.H) ? (unsigned)(irow_new * a.W * 4) : (unsigned)SX_OOB;
        const unsigned radd1 = (irow_new + 1 >= 0 && irow_new + 1 < a.H) ? (unsigned)((irow_new + 1) * a.W * 4) : (unsigned)SX_OOB;
#pragma unroll
        for (int it = 0; it < SX_ITEMS; ++it) {
            const unsigned radd = st_rr[it] ? radd1 : radd0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                raw[it][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (int)(st_off[it][e] + radd), 0, 0));
        }
    };
    auto stage_store = [&](const float (&raw)[SX_ITEMS][4], int irow_new) __attribute__((always_inline)) {
        const int s0 = (irow_new + 18) % SX_RING, s1 = (irow_new + 19) % SX_RING;
#pragma unroll
        for (int it = 0; it < SX_ITEMS; ++it) {
            const float* f = raw[it];
            float r[4], q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { r[e] = f[e] - sx_hi16(f[e]); q[e] = r[e] - sx_hi16(r[e]); }
            const int off = st_ok[it] ? (st_rr[it] ? s1 : s0) * slotb + st_l[it] : dump;
            const int pl = st_ok[it] ? rowb : 0;
            *reinterpret_cast<u32x2_t*>(smem + off) = u32x2_t{sx_pack_hi(f[0], f[1]), sx_pack_hi(f[2], f[3])};
            *reinterpret_cast<u32x2_t*>(smem + off + pl) = u32x2_t{sx_pack_hi(r[0], r[1]), sx_pack_hi(r[2], r[3])};
            *reinterpret_cast<u32x2_t*>(smem + off + 2 * pl) = u32x2_t{sx_pack_hi(q[0], q[1]), sx_pack_hi(q[2], q[3])};
        }
    };

    // ---- folded BatchNorm rows of this lane's accumulator registers: channel = mt*32 + (r & 3) + 8 (r >> 2) + 4 cl
    float sc[2][16], sh[2][16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * cl;
            sc[mt][r] = a.scale[ch]; sh[mt][r] = a.shift[ch];
        }

    // ---- prologue: input rows 2 cr_first - 3 ... + 4 (the seven of the first convolution row and the first new row of the second): all
    // requests first, one trip to memory
    {
        const int R0 = 2 * cr_first - 3;
        float pro[4][SX_ITEMS][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) stage_load(pro[k], R0 + 2 * k);
#pragma unroll
        for (int k = 0; k < 4; ++k) stage_store(pro[k], R0 + 2 * k);
        stage_load(raw, 2 * cr_first + 4);        // what the first convolution row writes between its matrix instructions (row + 4 again, row + 5)
    }
    u32x4_t af[3][2][3];                                                         // [ring slot][channel tile][plane]
    auto a_load = [&](int slot, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[slot][mt][p] = __builtin_amdgcn_raw_buffer_load_b128(wr, lane * 16, ((s * 2 + mt) * 3 + p) * 1024, 0);
    };
    a_load(0, 0);
    a_load(1, 1);
    __syncthreads();

    const int colb = (wave * 64 + nl) * 4;                                       // byte offset of this lane's first B fragment in a row
    // ---- pooling role: one pooled column, 32 channels.  Row cr's tile is pooled between the matrix instructions of row cr + 1.
    const int pc = tid & 127, pg = tid >> 7;
    const int pcc = min(pc, a.PWo - 1);
    const float* Tp = T + pg * 32 * a.LDT + 2 * pcc;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (long long)b * SX_CO * a.PH * a.PWo), 0, SX_CO * a.PH * a.PWo * 4, 0x00020000);
    const int chb = a.PH * a.PWo * 4;                                            // bytes of one pooled channel plane
    float vrun[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) vrun[i] = -__builtin_inff();
    // Channels i0 ... i0 + 3 of this thread's half: columns 2c-1, 2c, 2c+1 of a convolution row from the tile (pool_read), then rows 2r-1, 2r,
    // 2r+1 in vrun across calls (pool_use: an odd row closes pooled row row / 2 and opens the next one).  Read and use sit in DIFFERENT K-steps:
    // a wave alone on its SIMD issues in order, so a store that waits for its LDS read also stops the matrix instructions behind it
    // (measured: the pool cost 21 us of 107 with read -> maximum -> store back to back, nothing with either half removed).
    float pq[4][3];
    auto pool_read = [&](int i0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* t = Tp + (i0 + i) * a.LDT;
            const float2 m = *reinterpret_cast<const float2*>(t);
            pq[i][0] = t[-1]; pq[i][1] = m.x; pq[i][2] = m.y;       // (column 0: t[-1] is the pad of the previous tile row, replaced below)
        }
    };
    auto pool_use = [&](int i0, int row, bool valid) __attribute__((always_inline)) {
        const bool odd = row & 1;
        const int pr = row >> 1;
        const int yo = (valid && odd && pr >= r0 && pc < a.PWo) ? (pg * 32 * a.PH + pr) * a.PWo * 4 + pc * 4 : SX_OOB;       // nothing leaves otherwise
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float l = pcc > 0 ? pq[i][0] : -__builtin_inff();
            const float hm = valid ? fmaxf(fmaxf(l, pq[i][1]), pq[i][2]) : -__builtin_inff();      // (no branch: the first row of a workgroup has no predecessor)
            const float full = fmaxf(vrun[i0 + i], hm);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, full), yr, yo, (i0 + i) * chb, 0);
            vrun[i0 + i] = odd ? hm : full;
        }
    };
    auto pool = [&](int row, bool valid) __attribute__((always_inline)) {      // a whole row, not overlapped (waves without columns; the last row)
#pragma unroll
        for (int k = 0; k < 8; ++k) { pool_read(4 * k); pool_use(4 * k, row, valid); }
    };

#pragma unroll 1
    for (int cr = cr_first; cr <= cr_last; ++cr) {
        const int sb = (2 * cr - 3 + 18) % SX_RING;                              // ring slot of input row 2 cr - 3
        const bool have_prev = cr > cr_first;
        f32x16 acc[2][2];
        if (mma_wave) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][j][r] = 0.0f;
        }
        u32x4_t bf[2][2][3];
        auto b_read = [&](int set, int s) __attribute__((always_inline)) {
            const int p0 = 2 * s, p1 = min(2 * s + 1, SX_PAIRS - 1);
            int sl0 = sb + p0 % 7, sl1 = sb + p1 % 7;
            sl0 -= sl0 >= SX_RING ? SX_RING : 0; sl1 -= sl1 >= SX_RING ? SX_RING : 0;
            const int base = (cl ? sl1 * slotb + (p1 / 7) * 3 * rowb : sl0 * slotb + (p0 / 7) * 3 * rowb) + colb;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned* q = reinterpret_cast<const unsigned*>(ring + base + p * rowb + j * 128);
                    bf[set][j][p] = u32x4_t{q[0], q[1], q[2], q[3]};
                }
        };
        auto mma = [&](int slot, int set) __attribute__((always_inline)) {
#define DI2P_SX_PROD(QA, QB)                                                                                                            \
    _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int j = 0; j < 2; ++j)                                      \
        acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[slot][mt][QA]),                            \
                                                             __builtin_bit_cast(bf16x8_t, bf[set][j][QB]), acc[mt][j], 0, 0, 0);
            DI2P_SX_PROD(2, 0) DI2P_SX_PROD(1, 1) DI2P_SX_PROD(0, 2) DI2P_SX_PROD(1, 0) DI2P_SX_PROD(0, 1) DI2P_SX_PROD(0, 0)
#undef DI2P_SX_PROD
        };
        // The two input rows the NEXT convolution row adds (2 cr + 4, 2 cr + 5) were requested at the end of the previous row -- BEHIND the
        // weight requests of this row's first K-steps (memory returns a wave's loads in order: a wait for weights also waits for every older
        // load, and these come from HBM) -- and are written to their (dead) ring slots between the matrix instructions of K-step 6.  Behind
        // the last row they land in dead slots and are never read.  The previous row's tile is pooled between the matrix instructions, four
        // channels per K-step: read in one K-step, reduced and stored in the next.
        if (mma_wave) {
            b_read(0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < SX_KSTEPS; ++s) {
                if (s + 1 < SX_KSTEPS) b_read((s + 1) & 1, s + 1);
                if (s + 2 < SX_KSTEPS) a_load((s + 2) % 3, s + 2);
                mma(s % 3, s & 1);
                // pooling chunk k (four channels): read in K-step kr[k], used one K-step later (K-step 6 belongs to the staged rows)
                constexpr int kr[8] = {0, 1, 2, 3, 4, 7, 8, 9};
                if (s == 6) stage_store(raw, 2 * cr + 4);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (s == kr[k] + 1) pool_use(4 * k, cr - 1, have_prev);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (s == kr[k]) pool_read(4 * k);
                }
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           // one matrix instruction
                    if (i < 6 && s + 2 < SX_KSTEPS) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);           // one A request
                    else if (i >= 6 && i < 18 && s + 1 < SX_KSTEPS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one B read
                    if (s == 6) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                               // the split of the staged rows
                    const bool use_step = (s >= 1 && s <= 5) || (s >= 8 && s <= 10), read_step = s <= 4 || (s >= 7 && s <= 9);
                    if (use_step && i < 8) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                    // maxima of the chunk read one K-step ago
                    if (use_step && i >= 8 && i < 12) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);         // its four stores
                    if (read_step && i >= 18 && i < 22) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // the next chunk's tile reads
                }
                if (s == 6) __builtin_amdgcn_sched_group_barrier(0x200, 3 * SX_ITEMS, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // the first two K-steps of the next row: in flight across the barriers
            a_load(0, 0);
            a_load(1, 1);
            stage_load(raw, 2 * cr + 6);
        } else {
            stage_store(raw, 2 * cr + 4);
            stage_load(raw, 2 * cr + 6);
            pool(cr - 1, have_prev);
        }
        __syncthreads();                          // every thread has pooled the previous row: the tile is free
        if (mma_wave) {
            // ---- relu(scale * acc + shift) -> tile [channel][column]
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ch = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * cl;
                        T[ch * a.LDT + wave * 64 + j * 32 + nl] = fmaxf(acc[mt][j][r] * sc[mt][r] + sh[mt][r], 0.0f);
                    }
        }
        __syncthreads();                          // the tile of this row and the ring rows of the next one are complete
    }
    pool(cr_last, true);
}

}  // namespace

extern "C" long long di2p_stem_x3_packed_bytes(void) { return SX_WP_BYTES; }

// weight f32[64,3,7,7] (models/resnet.py:137 conv1) -> Wp (di2p_stem_x3_packed_bytes() bytes, 16-byte aligned): the split operand of di2p_stem_x3
extern "C" int di2p_stem_x3_pack(const float* weight, void* Wp, void* stream) {
    DI2P_CHECK_ARG(weight && Wp, "null pointer");
    DI2P_CHECK_ARG(((uintptr_t)Wp & 15) == 0, "packed weights must be 16-byte aligned");
    hipLaunchKernelGGL(stem_x3_pack_kernel, dim3(di2p_cdiv(SX_KSTEPS * 2 * 64, 256)), dim3(256), 0, (hipStream_t)stream, weight, (u32x4_t*)Wp);
    DI2P_RETURN_LAUNCH();
}

// 1 if di2p_stem_x3 runs an H x W image: H % 4 == 0, W % 128 == 0, W <= 512 (a workgroup spans the full width: four waves x 64 convolution columns)
// (includes the size limit of the kernel's buffer offsets -- a per-frame input below 2^30 bytes -- so that an image it admits never fails at the call)
extern "C" int di2p_stem_x3_supported(int H, int W) {
    return (H >= 4 && H % 4 == 0 && W >= 128 && W % 128 == 0 && W <= 512 && (long long)3 * H * W * 4 < (1ll << 30)) ? 1 : 0;
}

// y f32[B,64,H/4,W/4] = maxpool3x3/2/pad1( relu( scale * conv7x7/2/pad3(x f32[B,3,H,W]) + shift ) )
extern "C" int di2p_stem_x3(const float* x, const void* Wp, const float* scale, const float* shift, float* y, int B, int H, int W, void* stream) {
    DI2P_CHECK_ARG(x && Wp && scale && shift && y, "null pointer");
    DI2P_CHECK_ARG(B >= 0, "bad batch");
    DI2P_CHECK_ARG(di2p_stem_x3_supported(H, W), "needs H % 4 == 0, W % 128 == 0, W <= 512 and a per-frame input below 2^30 bytes");
    DI2P_CHECK_ARG(((uintptr_t)Wp & 15) == 0, "packed weights must be 16-byte aligned");
    if (B == 0) return 0;
    SxArgs a{};
    a.x = x; a.Wp = (const u32x4_t*)Wp; a.scale = scale; a.shift = shift; a.y = y;
    a.H = H; a.W = W; a.OH = H / 2; a.OW = W / 2; a.PH = H / 4; a.PWo = W / 4;
    // pooled rows per workgroup: a function of the image only (a frame's result must not depend on its batch -- here it would not, every
    // output has one summation order, but the launch shape stays batch independent like the other kernels'): eight workgroups per frame when
    // the image has the rows = one round of the chip at 32 frames, one halo row in 2 prw + 1
    a.prw = a.PH >= 8 ? di2p_cdiv(a.PH, 8) : 1;
    a.tiles = di2p_cdiv(a.PH, a.prw);
    a.LW = W + 8; a.LDT = a.OW + 4;
    const size_t lds = (size_t)SX_RING * 9 * a.LW * 2 + (size_t)SX_CO * a.LDT * 4 + 16;
    (void)hipFuncSetAttribute((const void*)stem_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(stem_x3_kernel, dim3(B * a.tiles), dim3(256), lds, (hipStream_t)stream, a);
    DI2P_RETURN_LAUNCH();
}
