// 3x3 convolutions (pad 1, stride 1 or 2) of the image branch as DIRECT implicit GEMMs on the bf16 matrix instructions with the EXACT
// three-way fp32 operand split of gemm.hip ("bf16x3": x = x1 + x2 + x3, three truncated bf16 terms = all 24 significand bits; six bf16
// products per fp32 product, smallest first, fp32 accumulation -- as accurate against fp64 as the fp32-MFMA kernels, 2.67x their
// matrix-pipe rate).  Replaces cuDNN's conv+BN+ReLU(+residual) of models/resnet.py:56-72,171-216 for the layers where a direct
// contraction on the bf16 pipe beats the fp32 Winograd kernel (winograd.hip): the 256- and 512-channel stages, the three stride-2
// layers -- whose 1x1 / stride-2 downsample branch (resnet.py:160-164) rides along as the centre tap of the same staged patch -- and,
// by measurement, the rest.
//
//   weights: split ONCE per checkpoint (di2p_bf16x3_pack of the tap-major [9 Cin][Cout] matrix: [K/8][Mp][3] x 8 bf16) and read straight
//            from L2 as MFMA A fragments (a wave owns its Cout rows: nothing to share through LDS), one K-step ahead;
//   activations: fp32 NCHW in memory.  A workgroup owns NSEG consecutive SEGMENTS (MF = 32 or 16 consecutive pixels of one output row) of
//            one frame and MT output channels.  Per chunk of CK input channels (one K-step: 16 on v_mfma_f32_32x32x16_bf16, 32 on
//            v_mfma_f32_16x16x32_bf16) the input PATCH under those segments (their rows plus the halo, zero padded) is split ONCE while it
//            is staged into LDS -- [channel group of 8][patch position][plane] x 16 bytes, so that one ds_read_b128 is a B fragment -- and
//            then serves all nine taps and every Cout tile of the workgroup: the inner loop is matrix instructions, 16-byte LDS reads and
//            one address add per read (the pointwise bf16x3 kernel re-splits per 128-row workgroup: 9 vector instructions per MFMA; here
//            about one).  A tap is a CONSTANT offset into the patch ((dy PW + dx) entries; for stride 2 the patch columns are stored
//            de-interleaved by parity so that consecutive output pixels stay consecutive entries): no im2col, no masks in the loop.
//   pipeline: the patch of chunk c+1 is fetched into registers at the head of chunk c and split + written to the OTHER LDS buffer between
//            the taps of chunk c (one barrier per chunk = per nine K-steps); where two patches do not fit in LDS (stride 2) a single
//            buffer is re-filled between two barriers.
//   epilogue: y = relu?(scale * acc + shift + residual), straight from the accumulators (a lane holds 4-row groups of one pixel column:
//            every store instruction writes whole 64/128-byte row pieces).
#include <stdint.h>

#include <type_traits>

#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MF> struct Mma;
template <> struct Mma<32> {
    typedef f32x16 acc_t;
    static constexpr int NACC = 16, KS = 16;
    static __device__ __forceinline__ acc_t mma(const u32x4_t& a, const u32x4_t& b, const acc_t& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    // accumulator register r of lane (column nl, k-group cl) = row:
    static __device__ __forceinline__ int row(int r, int cl) { return (r & 3) + 8 * (r >> 2) + 4 * cl; }
};
template <> struct Mma<16> {
    typedef f32x4 acc_t;
    static constexpr int NACC = 4, KS = 32;
    static __device__ __forceinline__ acc_t mma(const u32x4_t& a, const u32x4_t& b, const acc_t& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int r, int cl) { return r + 4 * cl; }
};

__device__ __forceinline__ float cx_hi16(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u); }
__device__ __forceinline__ unsigned cx_pack_hi(float x0, float x1) {      // bf16(x0) in the low half, bf16(x1) in the high half (truncation)
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
}
// eight consecutive input channels of one pixel -> the three planes of one LDS entry
__device__ __forceinline__ void cx_split8(const float (&f)[8], u32x4_t& p1, u32x4_t& p2, u32x4_t& p3) {
    float r[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { r[i] = f[i] - cx_hi16(f[i]); q[i] = r[i] - cx_hi16(r[i]); }
    p1 = u32x4_t{cx_pack_hi(f[0], f[1]), cx_pack_hi(f[2], f[3]), cx_pack_hi(f[4], f[5]), cx_pack_hi(f[6], f[7])};
    p2 = u32x4_t{cx_pack_hi(r[0], r[1]), cx_pack_hi(r[2], r[3]), cx_pack_hi(r[4], r[5]), cx_pack_hi(r[6], r[7])};
    p3 = u32x4_t{cx_pack_hi(q[0], q[1]), cx_pack_hi(q[2], q[3]), cx_pack_hi(q[4], q[5]), cx_pack_hi(q[6], q[7])};
}

#ifndef DI2P_CX_CLK
#define DI2P_CX_CLK 0
#endif
struct CxArgs {
    const float* x; const u32x4_t* Wp; const float* scale; const float* shift; const float* residual; float* y;
    const u32x4_t* Wp_ds; const float* scale_ds; const float* shift_ds; float* y_ds;      // fused 1x1 / stride-2 branch (DS instances)
    int Cin, H, W, Cout, OH, OW, Mp, Mp_ds;
    int spr, nseg, tiles_per_frame, n_mt;     // segments per output row, per frame; workgroup tiles per frame; Cout tiles
    int PW, PWH, PP, PPU;                     // patch row length, its even-column half (stride 2), entries per channel group (PRmax * PW = PPU, rounded up to 16)
    int relu;
};

// MF: matrix-instruction tile (32: 32x32x16, 16: 16x16x32); a wave owns TM x TN tiles (MF*TM output channels x TN segments), the workgroup
// WM x WN waves; STRIDE 1 / 2; DS: also the 1x1 / stride-2 convolution of the same input (centre tap, own weights and accumulators);
// DBUF: two patch buffers.  ITEMS: (patch position, channel group) pairs a thread stages per chunk.  PWT: the patch row length W + 2 as a
// compile-time constant (the nine tap offsets are then immediates of the LDS reads), 0: run-time (one address add per read).
// SA: the five small products of a K-step go to a SECOND accumulator set.  The bf16 matrix instructions align every product to the largest
// addend (normally the accumulator) and TRUNCATE what falls below its last bit (tools/probe_mfma_rounding.hip: 1 + 0.75 ulp -> 1 when the
// 0.75 ulp is a product of the same instruction): small products added to a large accumulator lose their low bits, with a bias.  Kept apart,
// they meet an accumulator 2^-8 times smaller.  Measured against fp64 on the four stride-1 shapes: rms error 3.6e-7 -> 1.5e-7 (K = 576) ...
// 9.1e-7 -> 3.6e-7 (K = 4608), below both fp32-MFMA kernels (4.2e-7 ... 6.3e-7); same speed.  Every shipped instance has it on.
#if DI2P_CX_CLK
// experiment: where a wave's lifetime goes (s_memtime stamps summed over all waves): [0] set-up + first patch + first weights up to the first
// barrier, [1] the chunks' taps, [2] the chunks' closing barriers, [3] epilogue, [7] waves
__device__ unsigned long long g_cx_clk[4096 * 4];          // one row per wave of the LAST launch (no atomics: they would be the epilogue)
extern "C" int di2p_cx_clk(unsigned long long* out, int nwaves) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cx_clk), (size_t)nwaves * 32) == hipSuccess ? 0 : -1;
}
#define CX_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); clk_[k] += now_ - last_; last_ = now_; } while (0)
#else
#define CX_STAMP(k) do { } while (0)
#endif
template <int MF, int TM, int TN, int WM, int WN, int STRIDE, bool DS, bool DBUF, int ITEMS, int PWT, bool SA>
__global__ __launch_bounds__(256, 1) void conv3x3_x3_kernel(const CxArgs a) {
    typedef Mma<MF> M;
    typedef typename M::acc_t acc_t;
    constexpr int KS = M::KS, CIGS = KS / 8, NSEG = TN * WN, MT = MF * TM * WM;
    static_assert(WM * WN == 4, "four waves");
    static_assert(!DS || STRIDE == 2, "the fused downsample branch belongs to the stride-2 layers");
    static_assert(!DBUF || ITEMS <= 6, "item k is fetched at tap k and written behind tap k + 3");
    static_assert(ITEMS <= 8, "one item per tap");
#if DI2P_CX_CLK
    unsigned long long clk_[4] = {0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
    const int PW = PWT ? PWT : a.PW, PWH = PWT ? (PWT + 1) / 2 : a.PWH;
    extern __shared__ __attribute__((aligned(16))) u32x4_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane % MF, cl = lane / MF;
    const int wm = wave / WN, wn = wave % WN;
    // Cout tile fastest: with eight of them (512 channels) an XCD's L2 holds ONE tile's split weights
    const int blk = blockIdx.x;
    const int mt = blk % a.n_mt, rest = blk / a.n_mt, tile = rest % a.tiles_per_frame, b = rest / a.tiles_per_frame;
    const int HW = a.H * a.W, OHW = a.OH * a.OW;
    const int s_first = tile * NSEG, r_lo = s_first / a.spr;
    const int s_last = min(s_first + NSEG, a.nseg) - 1, r_hi = s_last / a.spr;
    const int PR = STRIDE * (r_hi - r_lo) + 3, irow0 = STRIDE * r_lo - 1;
    const int BUF = CIGS * a.PP * 3 + 3;      // u32x4 entries per patch buffer; the last three are a dump for the items past the patch

    // ---- this wave's segments: B-fragment base (entries) and output offset
    int bbase[TN], ooff[TN];
    bool sval[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int s = s_first + wn * TN + j;
        sval[j] = s < a.nseg;
        const int sc = min(s, a.nseg - 1), srow = sc / a.spr, scol = (sc - srow * a.spr) * MF;
        bbase[j] = (cl * a.PP + (srow - r_lo) * STRIDE * PW + scol + nl) * 3;
        ooff[j] = srow * a.OW + scol + nl;
    }
    // ---- staging items of this thread: (patch position, channel group); position fastest across lanes (coalesced per channel plane)
    int g_off[ITEMS], l_off[ITEMS];
    bool g_ok[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int q = tid + 256 * it;
        const int cig = q / a.PP, pos = q - cig * a.PP, prow = pos / PW, pcol = pos - prow * PW;
        const int irow = irow0 + prow, icol = pcol - 1;
        const bool in_patch = cig < CIGS && pos < a.PPU;      // the padding entries behind a channel group's rows are nobody's (with the
                                                              // parity split of stride 2 their image would land in the NEXT group's first row)
        g_ok[it] = in_patch && prow < PR && irow >= 0 && irow < a.H && icol >= 0 && icol < a.W;
        g_off[it] = g_ok[it] ? (cig * 8 * HW + irow * a.W + icol) * 4 : 0;
        const int pc = STRIDE == 2 ? (pcol & 1) * PWH + (pcol >> 1) : pcol;
        l_off[it] = in_patch ? (cig * a.PP + prow * PW + pc) * 3 : BUF - 3;      // branch-free stores: surplus items land in the dump entry
    }
    const float* xb = a.x + (long long)b * a.Cin * HW;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, a.Cin * HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.Wp, 0, (int)min((long long)9 * (a.Cin / 8) * a.Mp * 48, 0x7fffffffll), 0x00020000);
    float raw[ITEMS][8];
    auto stage_load = [&](int it, int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            raw[it][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, g_off[it], (c * KS + i) * HW * 4, 0));
    };
    auto stage_store = [&](int it, int buf) __attribute__((always_inline)) {
        float f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = g_ok[it] ? raw[it][i] : 0.0f;
        u32x4_t p1, p2, p3;
        cx_split8(f, p1, p2, p3);
        u32x4_t* d = lds + buf * BUF + l_off[it];
        d[0] = p1; d[1] = p2; d[2] = p3;
    };
    // ---- weights: A fragments from memory; lane = (row nl of the tile, channel group cl of the K-step)
    int a_off[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_off[i] = (cl * a.Mp + mt * MT + (wm * TM + i) * MF + nl) * 48;
    const int kg_tap = a.Cin / 8;                 // channel groups per tap
    u32x4_t af[3][TM][3];                         // [ring slot][tile][plane]
    auto a_load = [&](int slot, int tap, int c) {
        const int so = (tap * kg_tap + c * CIGS) * a.Mp * 48;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[slot][i][p] = __builtin_amdgcn_raw_buffer_load_b128(wr, a_off[i], so + p * 16, 0);
    };
    acc_t acc[TM][TN], accs[SA ? TM : 1][SA ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) {
                acc[i][j][r] = 0.0f;
                if constexpr (SA) accs[i][j][r] = 0.0f;
            }
    // fused downsample branch: own descriptor, offsets, accumulators
    acc_t acc_ds[DS ? TM : 1][DS ? TN : 1];
    u32x4_t af_ds[DS ? TM : 1][3];
    int a_off_ds[DS ? TM : 1];
    __amdgpu_buffer_rsrc_t wr_ds = wr;
    if constexpr (DS) {
        wr_ds = __builtin_amdgcn_make_buffer_rsrc((void*)a.Wp_ds, 0, (int)min((long long)(a.Cin / 8) * a.Mp_ds * 48, 0x7fffffffll), 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            a_off_ds[i] = (cl * a.Mp_ds + mt * MT + (wm * TM + i) * MF + nl) * 48;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < M::NACC; ++r) acc_ds[i][j][r] = 0.0f;
        }
    }
    auto a_load_ds = [&](int c) {
        if constexpr (DS) {
            const int so = c * CIGS * a.Mp_ds * 48;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) af_ds[i][p] = __builtin_amdgcn_raw_buffer_load_b128(wr_ds, a_off_ds[i], so + p * 16, 0);
        }
    };
    const int NC = a.Cin / KS;

    // B fragments of one tap into register set `set` (two sets: the reads of tap t+1 are issued among the matrix instructions of tap t)
    u32x4_t bf[2][TN][3];
    int bcur[TN];                                 // bbase + the current patch buffer
    auto b_read = [&](int set, int tap) __attribute__((always_inline)) {
        const int dy = tap / 3, dx = tap - 3 * dy;
        const int toff = (STRIDE == 2 ? dy * PW + (dx & 1) * PWH + (dx >> 1) : dy * PW + dx) * 3;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[set][j][p] = lds[bcur[j] + toff + p];
    };
    // one tap of one chunk: six products per tile, smallest terms first; consecutive matrix instructions write different accumulators
    auto tap_mma = [&](int slot, int set, int tap) __attribute__((always_inline)) {
#define DI2P_CX_PROD(ACC, QA, QB)                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                                       \
        ACC[i][j] = M::mma(af[slot][i][QA], bf[set][j][QB], ACC[i][j]);
        if constexpr (SA) {
            DI2P_CX_PROD(accs, 2, 0) DI2P_CX_PROD(accs, 1, 1) DI2P_CX_PROD(accs, 0, 2) DI2P_CX_PROD(accs, 1, 0) DI2P_CX_PROD(accs, 0, 1)
        } else {
            DI2P_CX_PROD(acc, 2, 0) DI2P_CX_PROD(acc, 1, 1) DI2P_CX_PROD(acc, 0, 2) DI2P_CX_PROD(acc, 1, 0) DI2P_CX_PROD(acc, 0, 1)
        }
        DI2P_CX_PROD(acc, 0, 0)
#undef DI2P_CX_PROD
        if constexpr (DS) {
            if (tap == 4) {
#define DI2P_CX_PROD(QA, QB)                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                                       \
        acc_ds[i][j] = M::mma(af_ds[i][QA], bf[set][j][QB], acc_ds[i][j]);
                DI2P_CX_PROD(2, 0) DI2P_CX_PROD(1, 1) DI2P_CX_PROD(0, 2) DI2P_CX_PROD(1, 0) DI2P_CX_PROD(0, 1) DI2P_CX_PROD(0, 0)
#undef DI2P_CX_PROD
            }
        }
    };
    // One chunk; STAGE: the next chunk's patch is fetched and written while this one is multiplied.  A wave alone on its SIMD issues in
    // order: whatever is not issued BETWEEN two matrix instructions idles the matrix pipe.  Between two scheduling fences a tap is therefore
    // laid out (sched_group_barrier) as NM x { one matrix instruction of tap t ; one operand request of tap t+1 -- first the A fragments from
    // memory into the next ring slot, then the B fragments from LDS into the other register set ; a few vector instructions of the split
    // of one staged item }, then that item's LDS stores and last the eight loads of the item that is fetched in this tap (behind the
    // A requests: the wait for A at the next tap then leaves them in flight; they are first used three taps later).
    auto chunk = [&](int c, auto stage_tag) __attribute__((always_inline)) {
        constexpr bool STAGE = decltype(stage_tag)::value;
        const int buf = DBUF ? (c & 1) : 0;
#pragma unroll
        for (int j = 0; j < TN; ++j) bcur[j] = bbase[j] + buf * BUF;
        a_load_ds(c);
        b_read(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            constexpr int NM = TM * TN * 6, NB = 3 * TN, NA = 3 * TM;
            // the B fragments of the next K-step and the A fragments of the one after it (ring of three slots, 9 % 3 == 0: slot = tap % 3 in
            // every chunk; past the end: a valid, unused address).  Two K-steps ahead because memory returns a wave's loads IN ORDER: the wait
            // for these weights also waits for every staged-patch load issued before them, which comes from HBM
            if (tap < 8) b_read((tap + 1) & 1, tap + 1);
            if (tap < 7) a_load((tap + 2) % 3, tap + 2, c);
            else a_load((tap + 2) % 3, tap - 7, min(c + 1, NC - 1));
            tap_mma(tap % 3, tap & 1, tap);
            const bool stores = STAGE && DBUF && tap >= 3 && tap - 3 < ITEMS, loads = STAGE && tap < ITEMS;
            if (stores) stage_store(tap - 3, buf ^ 1);
            if (loads) stage_load(tap, c + 1);
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    // one matrix instruction
                if (i < NA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                        // one A request
                else if (i < NA + NB && tap < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one B request
                if (loads && i >= NA && i < NA + 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // one load of the item fetched in this tap
                if (stores) __builtin_amdgcn_sched_group_barrier(0x002, (56 + NM - 1) / NM, 0);       // the split: ~56 vector instructions
            }
            if (stores) __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        CX_STAMP(1);
        if constexpr (STAGE && !DBUF) {
            __syncthreads();                  // every wave has read the patch of chunk c
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) stage_store(it, 0);
        }
        __syncthreads();
        CX_STAMP(2);
    };
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) stage_load(it, 0);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) stage_store(it, 0);
    a_load(0, 0, 0);
    a_load(1, 1, 0);
    __syncthreads();
    CX_STAMP(0);
    for (int c = 0; c + 1 < NC; ++c) chunk(c, std::true_type{});
    chunk(NC - 1, std::false_type{});

    // ---- epilogue
    if constexpr (SA) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] += accs[i][j];
    }
    // Every load of the epilogue (folded BatchNorm rows, residual) is requested before the first result is formed, and nothing in it
    // branches: hipcc otherwise waits for each residual value (and the store before it) in turn -- eighty memory round trips per wave.
    // Buffer addressing drops what must not be written (segments past the frame, channels past Cout): their offset is out of range.
    auto store = [&](const acc_t (&ac)[TM][TN], const float* scale, const float* shift, const float* res, float* y, bool relu,
                     auto res_tag) __attribute__((always_inline)) {
        constexpr bool RES = decltype(res_tag)::value;
        constexpr int OOB = 0x40000000;
        const int bytes = a.Cout * OHW * 4;
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (long long)b * a.Cout * OHW), 0, bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)((RES ? res : y) + (long long)b * a.Cout * OHW), 0, bytes, 0x00020000);
        int so[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) so[j] = sval[j] ? ooff[j] * 4 : OOB;
        float sc[TM][M::NACC], sh[TM][M::NACC], rv[TM][TN][RES ? M::NACC : 1];
        int co_off[TM][M::NACC];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) {
                const int co = mt * MT + (wm * TM + i) * MF + M::row(r, cl), cc = min(co, a.Cout - 1);
                sc[i][r] = scale[cc]; sh[i][r] = shift[cc];
                co_off[i][r] = co < a.Cout ? co * OHW * 4 : OOB;
                if constexpr (RES) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        rv[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (int)((unsigned)co_off[i][r] + (unsigned)so[j]), 0, 0));
                }
            }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < M::NACC; ++r)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float v = ac[i][j][r] * sc[i][r] + sh[i][r];
                    if constexpr (RES) v += rv[i][j][r];
                    v = relu ? fmaxf(v, 0.0f) : v;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, (int)((unsigned)co_off[i][r] + (unsigned)so[j]), 0, 0);
                }
    };
    if (a.residual) store(acc, a.scale, a.shift, a.residual, a.y, a.relu != 0, std::true_type{});
    else store(acc, a.scale, a.shift, a.residual, a.y, a.relu != 0, std::false_type{});
    if constexpr (DS) store(acc_ds, a.scale_ds, a.shift_ds, nullptr, a.y_ds, false, std::false_type{});
#if DI2P_CX_CLK
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CX_STAMP(3);
    if (lane == 0 && blockIdx.x * 4 + wave < 4096)
        for (int k = 0; k < 4; ++k) g_cx_clk[(blockIdx.x * 4 + wave) * 4 + k] = clk_[k];
#endif
}

struct CxCfg { int MF, TM, TN, WM, WN; };
constexpr CxCfg kCfgs[4] = {{32, 1, 5, 2, 2}, {32, 1, 5, 4, 1}, {16, 2, 5, 4, 1}, {16, 1, 5, 4, 1}};
constexpr int CX_LDS_MAX = 160 * 1024;

struct CxPlan {
    int cfg = -1, items = 0, dbuf = 0, tiles_per_frame = 0, n_mt = 0, PW = 0, PWH = 0, PP = 0, PPU = 0, spr = 0, nseg = 0;
    long long lds = 0, cost = 0;
};

// geometry of configuration `ci` on this layer; cfg = -1 when it cannot run (shape or LDS)
CxPlan cx_plan(int ci, int B, int Cin, int H, int W, int Cout, int stride) {
    CxPlan p;
    const CxCfg& c = kCfgs[ci];
    const int KS = c.MF == 32 ? 16 : 32, CIGS = KS / 8, NSEG = c.TN * c.WN, MT = c.MF * c.TM * c.WM;
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    if (OW % c.MF != 0 || Cin % KS != 0) return p;
    p.spr = OW / c.MF; p.nseg = OH * p.spr;
    p.tiles_per_frame = di2p_cdiv(p.nseg, NSEG);
    p.n_mt = di2p_cdiv(Cout, MT);
    int rows = 1;
    for (int t = 0; t < p.tiles_per_frame; ++t) {
        const int s0 = t * NSEG, s1 = (s0 + NSEG < p.nseg ? s0 + NSEG : p.nseg) - 1;
        const int r = s1 / p.spr - s0 / p.spr + 1;
        rows = r > rows ? r : rows;
    }
    const int PRmax = stride * (rows - 1) + 3;
    // entries per channel group, a multiple of 16: with 16-pixel segments the four channel groups of a wave's ds_read_b128 then fall on
    // disjoint bank sets (48-byte entries: 16 of them are 3 x 256 bytes)
    p.PW = W + 2; p.PWH = (p.PW + 1) / 2; p.PPU = PRmax * p.PW; p.PP = (p.PPU + 15) / 16 * 16;
    const long long buf = (long long)CIGS * p.PP * 48 + 48;
    if (2 * buf <= CX_LDS_MAX - 1024) { p.dbuf = 1; p.lds = 2 * buf; }
    else if (buf <= CX_LDS_MAX - 1024) { p.dbuf = 0; p.lds = buf; }
    else return p;
    p.items = di2p_cdiv((long long)CIGS * p.PP, 256);
    if (p.items > 8) return p;
    // the choice must not depend on the batch a frame is in (different blockings add in different orders: a frame's result would change
    // with the batch size): priced for the nominal 32-frame step whatever B is
    (void)B;
    const long long wgs = 32ll * p.tiles_per_frame * p.n_mt;
    const long long per_wg = (long long)(Cin / KS) * 9 * c.TM * c.TN * 6 * (c.MF == 32 ? 32 : 16) + (long long)(Cin / KS) * p.items * 400 * (p.dbuf ? 1 : 4);
    p.cost = di2p_cdiv(wgs, di2p_cu_count()) * per_wg;
    p.cfg = ci;
    return p;
}

template <int MF, int TM, int TN, int WM, int WN, int STRIDE, bool DS, bool DBUF, int ITEMS, int PWT, bool SA>
void cx_launch_one(const CxArgs& a, int grid, size_t lds, hipStream_t st) {
    auto k = conv3x3_x3_kernel<MF, TM, TN, WM, WN, STRIDE, DS, DBUF, ITEMS, PWT, SA>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
}

// The instantiated kernels.  Each row: configuration, stride, double-buffered patch, ITEMS, PWT (0 = any patch row length; a plan that
// needs fewer items runs the next larger instance of its row length, the surplus items are masked).  Stride-2 instances carry the fused
// downsample branch.  The PWT != 0 rows are the seven layer shapes of ResNet-34 at 160 x 512 (in the configuration the plan picks for them).
#define DI2P_CX_INSTANCES(X)                                                                                                            \
    X(0, 32, 1, 5, 2, 2, 1, 1, 6, 130) X(1, 32, 1, 5, 4, 1, 1, 1, 3, 66) X(2, 16, 2, 5, 4, 1, 1, 1, 3, 34) X(3, 16, 1, 5, 4, 1, 1, 1, 2, 18) \
    X(1, 32, 1, 5, 4, 1, 2, 0, 8, 130) X(2, 16, 2, 5, 4, 1, 2, 0, 8, 66) X(3, 16, 1, 5, 4, 1, 2, 1, 6, 34)                                \
    X(0, 32, 1, 5, 2, 2, 1, 1, 6, 0) X(1, 32, 1, 5, 4, 1, 1, 1, 4, 0) X(2, 16, 2, 5, 4, 1, 1, 1, 4, 0) X(3, 16, 1, 5, 4, 1, 1, 1, 4, 0)     \
    X(1, 32, 1, 5, 4, 1, 2, 0, 8, 0) X(3, 16, 1, 5, 4, 1, 2, 0, 8, 0)

struct CxInst { int cfg, stride, dbuf, items, pwt; };
#define DI2P_CX_ROW(CFG, MF, TM, TN, WM, WN, STRIDE, DBUF, ITEMS, PWT) {CFG, STRIDE, DBUF, ITEMS, PWT},
const CxInst kInst[] = {DI2P_CX_INSTANCES(DI2P_CX_ROW)};
#undef DI2P_CX_ROW

// the instance a plan runs on (its index in kInst, -1: none): exact row length first, then the run-time one; fewest items that suffice
int cx_find_instance(int cfg, int stride, int dbuf, int need, int pw) {
    int best = -1;
    for (int pass = 0; pass < 2 && best < 0; ++pass)
        for (int i = 0; i < (int)(sizeof(kInst) / sizeof(kInst[0])); ++i) {
            const CxInst& k = kInst[i];
            if (k.cfg != cfg || k.stride != stride || k.dbuf != dbuf || k.items < need || k.pwt != (pass == 0 ? pw : 0)) continue;
            if (best < 0 || k.items < kInst[best].items) best = i;
        }
    return best;
}

bool cx_launch(int inst, const CxPlan& p, const CxArgs& a, int grid, hipStream_t st) {
    int i = 0;
#define DI2P_CX_CASE(CFG, MF, TM, TN, WM, WN, STRIDE, DBUF, ITEMS, PWT)                                                                \
    if (inst == i++) {                                                                                                                 \
        cx_launch_one<MF, TM, TN, WM, WN, STRIDE, STRIDE == 2, DBUF != 0, ITEMS, PWT, true>(a, grid, (size_t)p.lds, st);              \
        return true;                                                                                                                    \
    }
    DI2P_CX_INSTANCES(DI2P_CX_CASE)
#undef DI2P_CX_CASE
    return false;
}

// the cheapest runnable plan of a layer (cfg = -1: none); `force` >= 0 restricts it to one configuration
CxPlan cx_best(int B, int Cin, int H, int W, int Cout, int stride, long long force, int* inst_out) {
    CxPlan best;
    for (int ci = 0; ci < 4; ++ci) {
        if (force >= 0 && force < 4 && force != ci) continue;
        if (force >= 4 && ((force >> 2) & (1 << ci)) == 0) continue;      // force = 4 * (bit mask of admitted configurations): experiments
        CxPlan p = cx_plan(ci, B, Cin, H, W, Cout, stride);
        if (p.cfg < 0) continue;
        int inst = cx_find_instance(ci, stride, p.dbuf, p.items, p.PW);
        if (inst < 0 && p.dbuf) {                  // no double-buffered instance: the single-buffered one of this configuration
            p.dbuf = 0; p.lds /= 2;
            inst = cx_find_instance(ci, stride, 0, p.items, p.PW);
        }
        if (inst < 0) continue;
        p.items = kInst[inst].items;
        if (best.cfg < 0 || p.cost < best.cost) { best = p; if (inst_out) *inst_out = inst; }
    }
    return best;
}

// The size limits of the kernels' 31-bit buffer offsets and of the grid: ONE predicate for `supported()` and for the launch entry (a layer
// that `supported()` admits must not fail at the call: the host layers fall back to the fp32 kernels on `supported() == 0` only).
const char* cx_size_limit(int B, int Cin, int H, int W, int Cout, const CxPlan* plan) {
    if ((long long)Cout * H * W * 4 >= (1ll << 30)) return "per-frame output must stay below 2^30 bytes";
    if ((long long)Cin * H * W * 4 >= (1ll << 31) || (long long)9 * (Cin / 8) * (di2p_cdiv(Cout, 128) * 128) * 48 >= (1ll << 31))
        return "per-frame input and the packed weights must fit 31-bit byte offsets";
    if (plan && (long long)B * plan->tiles_per_frame * plan->n_mt >= (1ll << 31)) return "too many workgroups";
    return nullptr;
}

}  // namespace

// 1 if di2p_conv3x3_x3 can run this layer (some tile configuration fits its shape and the LDS, and the sizes fit the kernels' offsets), else 0.
extern "C" int di2p_conv3x3_x3_supported(int B, int Cin, int H, int W, int Cout, int stride) {
    if (B < 1 || Cin < 16 || H < 1 || W < 1 || Cout < 1 || (stride != 1 && stride != 2)) return 0;
    if (stride == 2 && (H % 2 || W % 2)) return 0;
    const CxPlan best = cx_best(B, Cin, H, W, Cout, stride, di2p_opt(DI2P_OPT_CONV_X3_CFG), nullptr);
    return best.cfg >= 0 && cx_size_limit(B, Cin, H, W, Cout, &best) == nullptr ? 1 : 0;
}

// y f32[B,Cout,OH,OW] = relu?( scale * conv3x3(x f32[B,Cin,H,W]; pad 1, stride 1|2) + shift + residual ), weights Wp = di2p_bf16x3_pack of
// the tap-major matrix Wt[(kh*3+kw)*Cin + ci][Cout].  Optional second output of the SAME input (stride 2 only): y_ds f32[B,Cout,OH,OW] =
// scale_ds * conv1x1/stride-2(x) + shift_ds with Wp_ds = di2p_bf16x3_pack of Wt_ds[Cin][Cout] (the BasicBlock's downsample branch,
// models/resnet.py:160-164,62-63).
extern "C" int di2p_conv3x3_x3(const float* x, const void* Wp, const float* scale, const float* shift, const float* residual, float* y, int B,
                               int Cin, int H, int W, int Cout, int stride, int relu, const void* Wp_ds, const float* scale_ds,
                               const float* shift_ds, float* y_ds, void* stream) {
    DI2P_CHECK_ARG(x && Wp && scale && shift && y, "null pointer");
    DI2P_CHECK_ARG(B >= 0 && Cin >= 16 && H >= 1 && W >= 1 && Cout >= 1 && (stride == 1 || stride == 2), "bad shape");
    DI2P_CHECK_ARG(stride == 1 || (H % 2 == 0 && W % 2 == 0), "stride 2 needs even H and W");
    DI2P_CHECK_ARG(((uintptr_t)Wp & 15) == 0 && ((uintptr_t)Wp_ds & 15) == 0, "packed weights must be 16-byte aligned");
    const bool ds = Wp_ds != nullptr;
    DI2P_CHECK_ARG(ds == (stride == 2), "stride 2 runs WITH the fused 1x1 / stride-2 branch of the same input (and only stride 2 has one)");
    DI2P_CHECK_ARG(!ds || (scale_ds && shift_ds && y_ds), "the fused 1x1 branch needs its scale / shift / output");
    {
        const char* why = cx_size_limit(B, Cin, H, W, Cout, nullptr);
        DI2P_CHECK_ARG(why == nullptr, why);
    }
    if (B == 0) return 0;
    int inst = -1;
    const CxPlan best = cx_best(B, Cin, H, W, Cout, stride, di2p_opt(DI2P_OPT_CONV_X3_CFG), &inst);
    DI2P_CHECK_ARG(best.cfg >= 0, "no kernel instance fits this shape (needs OW % 32 == 0 and Cin % 16 == 0, or OW % 16 == 0 and Cin % 32 == 0, and a patch that fits the LDS)");
    CxArgs a{};
    a.x = x; a.Wp = (const u32x4_t*)Wp; a.scale = scale; a.shift = shift; a.residual = residual; a.y = y;
    a.Wp_ds = (const u32x4_t*)Wp_ds; a.scale_ds = scale_ds; a.shift_ds = shift_ds; a.y_ds = y_ds;
    a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = (H - 1) / stride + 1; a.OW = (W - 1) / stride + 1;
    a.Mp = di2p_cdiv(Cout, 128) * 128; a.Mp_ds = a.Mp;
    a.spr = best.spr; a.nseg = best.nseg; a.tiles_per_frame = best.tiles_per_frame; a.n_mt = best.n_mt;
    a.PW = best.PW; a.PWH = best.PWH; a.PP = best.PP; a.PPU = best.PPU; a.relu = relu;
    const long long grid = (long long)B * best.tiles_per_frame * best.n_mt;
    DI2P_CHECK_ARG(cx_size_limit(B, Cin, H, W, Cout, &best) == nullptr, "too many workgroups");
    hipStream_t st = (hipStream_t)stream;
    const bool ok = cx_launch(inst, best, a, (int)grid, st);
    DI2P_CHECK_ARG(ok, "internal: no kernel instance for the plan");
    DI2P_RETURN_LAUNCH();
}
