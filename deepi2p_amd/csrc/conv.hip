// ResNet-34 image branch kernels (models/resnet.py:56-72,125-216) for gfx950.
//
// conv + BN(eval) + [residual] + [ReLU] as ONE implicit-GEMM kernel on fp32 MFMA:
//   M = Cout, N = B*OH*OW (output pixels of the whole batch, so /32 maps still fill the chip),
//   K = Cin*KH*KW, tap-major (kh,kw,ci) for the 3x3 / 1x1 layers (one filter tap per K-step), the weight's own
//   (ci,kh,kw) order for the 7x7 stem.  The B operand is the im2col view of the NCHW input gathered on the fly (never
//   materialised): for a fixed k the lanes of an MFMA operand read consecutive output pixels = consecutive input
//   addresses (stride 1) of one input row, straight from the reference's own layout; 16-byte loads, padding applied
//   at LDS-store time, XCD-aware tile order, deterministic split-K for the layers with few output pixels.
#include "mfma_tile.h"

#include <stdlib.h>

namespace {

struct LoaderWtC {
    const float* Wt;  // [K][Cout]
    int K, M;
    __device__ __forceinline__ float load(int k, int m) const { return (k < K && m < M) ? Wt[k * M + m] : 0.0f; }   // K*M < 2^31 (host-checked)
};

struct LoaderIm2col {
    const float* x;
    int Cin, H, W, OH, OW, KH, KW, stride, pad, K, Ntot;
    const float* xb;
    int ih0, iw0;
    bool valid;
    __device__ __forceinline__ void column(int j) {
        valid = j < Ntot;
        const int jj = valid ? j : 0;
        const int opix = OH * OW;
        const int b = jj / opix, pix = jj - b * opix;
        const int oh = pix / OW, ow = pix - oh * OW;
        xb = x + (long long)b * Cin * H * W;
        ih0 = oh * stride - pad;
        iw0 = ow * stride - pad;
    }
    __device__ __forceinline__ void begin_tile(int) {}
    __device__ __forceinline__ float load(int k) const {
        k = __builtin_amdgcn_readfirstlane(k);
        if (!valid || k >= K) return 0.0f;
        int ci, kh, kw;
        if (KH == 7 && KW == 7) {            // the ResNet stem: divisions by constants
            ci = k / 49; const int rem = k - ci * 49; kh = rem / 7; kw = rem - kh * 7;
        } else if (KH == 3 && KW == 3) {
            ci = k / 9; const int rem = k - ci * 9; kh = rem / 3; kw = rem - kh * 3;
        } else {
            const int khw = KH * KW;
            ci = k / khw; const int rem = k - ci * khw; kh = rem / KW; kw = rem - kh * KW;
        }
        const int ih = ih0 + kh, iw = iw0 + kw;
        if ((unsigned)ih >= (unsigned)H || (unsigned)iw >= (unsigned)W) return 0.0f;
        return xb[((long long)ci * H + ih) * W + iw];
    }
};

// Tap-major K order: k' = (kh*KW + kw)*Cin + ci with Cin % 16 == 0, so one 16-row K-step lies inside ONE filter
// tap: the tap decode, the bounds test and the base address are computed once per K-step per lane, and the
// staged rows are constant-stride (H*W) loads.  (The generic loader above pays two integer divisions per
// staged element, which made the kernel VALU-bound at ~40 TF.)
struct LoaderIm2colTap {
    const float* x;
    int Cin, H, W, OH, OW, KH, KW, stride, pad, K, Ntot;
    const float* xb;
    const float* tile_ptr;   // xb + (ci0*H + ih)*W + iw of the current K-step (nullptr: out of bounds)
    int ih0, iw0, HW;
    int ci0, kh, kw;         // wave-uniform walk over the taps
    bool valid;
    __device__ __forceinline__ void column(int j) {
        valid = j < Ntot;
        const int jj = valid ? j : 0;
        const int opix = OH * OW;
        const int b = jj / opix, pix = jj - b * opix;
        const int oh = pix / OW, ow = pix - oh * OW;
        HW = H * W;
        xb = x + (long long)b * Cin * HW;
        ih0 = oh * stride - pad;
        iw0 = ow * stride - pad;
        ci0 = -bk; kh = 0; kw = 0;
        tile_ptr = nullptr;
    }
    int bk;                  // K-step (16 or 32); Cin % bk == 0
    __device__ __forceinline__ void begin_tile(int) {
        ci0 += bk;
        if (ci0 >= Cin) { ci0 = 0; if (++kw == KW) { kw = 0; ++kh; } }
        const int ih = ih0 + kh, iw = iw0 + kw;
        const bool ok = valid && kh < KH && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        tile_ptr = ok ? xb + (ci0 * H + ih) * W + iw : nullptr;
    }
    __device__ __forceinline__ float load(int k) const { return tile_ptr ? tile_ptr[(k & (bk - 1)) * HW] : 0.0f; }
};

// Vector stager (16-byte loads), tap-major weights.  Four consecutive output pixels of one output row per lane
// (OW % 4 == 0).  All loads are UNCONDITIONAL from clamped addresses -- no control flow in the K-loop, so the loads of
// K-step t+1 stay in flight under the MFMAs of step t -- and padding is applied by fix() at LDS-store time:
//   stride 1 (pad <= 1): ONE dword-aligned 16-byte load from the 4-float window clamped into the input row; at the
//             left/right image border the window is off by one, fix() shifts the components and zeroes the pad;
//   stride 2: four scalar loads at clamped columns, fix() zeroes the out-of-range ones.
// A row outside the image is read from the clamped row and zeroed as a whole.
template <int STRIDE>
struct LoaderIm2colTap4 {
    const float* x;
    int Cin, H, W, OH, OW, KH, KW, pad, Ntot;
    const float* xb;
    const float* tile_ptr;   // channel ci0 of the (clamped) input row, at the clamped first column (stride 1) / column 0 (stride 2)
    int ih0, iw0, HW;
    int ci0, kh, kw, bk;
    int c0, c1, c2, c3;      // stride 2: clamped columns
    int shift;               // stride 1: wanted first column - clamped first column, in {-1, 0, +1}
    unsigned okmask;         // bit i: element i is inside the image
    __device__ __forceinline__ void column4(int j) {
        const int jj = j < Ntot ? j : 0;       // out-of-range groups compute garbage that the epilogue drops
        const int opix = OH * OW;
        const int b = jj / opix, pix = jj - b * opix;
        const int oh = pix / OW, ow = pix - oh * OW;
        HW = H * W;
        xb = x + (long long)b * Cin * HW;
        ih0 = oh * STRIDE - pad;
        iw0 = ow * STRIDE - pad;
        ci0 = -bk; kh = 0; kw = 0;
        if (pending_seek > 0) seek(pending_seek);
        tile_ptr = xb; shift = 0; okmask = 0; c0 = c1 = c2 = c3 = 0;
    }
    int pending_seek = 0;
    // position the tap walk so that the next begin_tile() lands on K-step t0 (split-K blocks start mid-way)
    __device__ __forceinline__ void seek(int t0) {
        const int k0 = t0 * bk, tap = k0 / Cin;
        ci0 = k0 - tap * Cin - bk;
        kh = tap / KW; kw = tap - kh * KW;
    }
    __device__ __forceinline__ void begin_tile(int) {
        ci0 += bk;
        if (ci0 >= Cin) { ci0 = 0; if (++kw == KW) { kw = 0; ++kh; } }
        const int ih = ih0 + kh, iw = iw0 + kw;
        const bool row_ok = (unsigned)ih < (unsigned)H;
        const int ihc = min(max(ih, 0), H - 1);
        const float* row = xb + (ci0 * H + ihc) * W;
        if (STRIDE == 1) {
            const int cs = min(max(iw, 0), W - 4);
            shift = iw - cs;
            tile_ptr = row + cs;
            okmask = row_ok ? 0xfu : 0u;
        } else {
            c0 = min(max(iw, 0), W - 1); c1 = min(max(iw + STRIDE, 0), W - 1);
            c2 = min(max(iw + 2 * STRIDE, 0), W - 1); c3 = min(max(iw + 3 * STRIDE, 0), W - 1);
            tile_ptr = row;
            okmask = !row_ok ? 0u : ((unsigned)iw < (unsigned)W ? 1u : 0u) | ((unsigned)(iw + STRIDE) < (unsigned)W ? 2u : 0u) |
                                    ((unsigned)(iw + 2 * STRIDE) < (unsigned)W ? 4u : 0u) | ((unsigned)(iw + 3 * STRIDE) < (unsigned)W ? 8u : 0u);
        }
    }
    __device__ __forceinline__ float4 load4(int k) const {
        const float* r = tile_ptr + (k & (bk - 1)) * HW;
        if (STRIDE == 1) { const F4u t = *reinterpret_cast<const F4u*>(r); return make_float4(t.x, t.y, t.z, t.w); }
        return make_float4(r[c0], r[c1], r[c2], r[c3]);
    }
    struct Info { int shift; unsigned okmask; };
    __device__ __forceinline__ Info info() const { return Info{shift, okmask}; }
    __device__ __forceinline__ void fix(float4& v, int, const Info& in) const {
        if (STRIDE == 1) {
            const float4 t = v;
            const bool l = in.shift < 0, r = in.shift > 0, ok = in.okmask != 0;
            v.x = !ok || l ? 0.0f : (r ? t.y : t.x);
            v.y = !ok ? 0.0f : (l ? t.x : (r ? t.z : t.y));
            v.z = !ok ? 0.0f : (l ? t.y : (r ? t.w : t.z));
            v.w = !ok || r ? 0.0f : (l ? t.z : t.w);
        } else {
            v.x = (in.okmask & 1u) ? v.x : 0.0f; v.y = (in.okmask & 2u) ? v.y : 0.0f;
            v.z = (in.okmask & 4u) ? v.z : 0.0f; v.w = (in.okmask & 8u) ? v.w : 0.0f;
        }
    }
    __device__ __forceinline__ void fix(float4& v, int) const {
        if (STRIDE == 1) {
            const float4 t = v;
            const bool l = shift < 0, r = shift > 0, ok = okmask != 0;
            v.x = !ok || l ? 0.0f : (r ? t.y : t.x);
            v.y = !ok ? 0.0f : (l ? t.x : (r ? t.z : t.y));
            v.z = !ok ? 0.0f : (l ? t.y : (r ? t.w : t.z));
            v.w = !ok || r ? 0.0f : (l ? t.z : t.w);
        } else {
            v.x = (okmask & 1u) ? v.x : 0.0f; v.y = (okmask & 2u) ? v.y : 0.0f;
            v.z = (okmask & 4u) ? v.z : 0.0f; v.w = (okmask & 8u) ? v.w : 0.0f;
        }
    }
};

// Stride 2, WINDOW form (round 4).  The four output pixels ow .. ow + 3 (ow % 4 == 0) of a lane read the input columns 2 ow - pad + kw + {0, 2, 4, 6}:
// with W == 2 OW and (pad, KW) = (1, 3) or (0, 1) they lie in the 8-float window [2 ow, 2 ow + 8) of the input row plus, for the leftmost tap of a
// padded 3x3, the one element to its left.  Two aligned 16-byte loads and one dword per staged row -- whole cache lines across the wave -- instead of
// four dword loads with a 32-byte lane pitch (half of every line fetched and dropped, twice the address work in the texture path); finish() picks the
// four values by the tap's offset o = kw - pad in {-1, 0, +1}.  Same values into LDS as LoaderIm2colTap4<2>: results are bit-identical.
struct LoaderIm2colTap4W2 {
    struct Raw { float4 a, b, c; };          // c.x: the element left of the window (three float4s: no padding bytes for the register promotion to trip over)
    const float* x;
    int Cin, H, W, OH, OW, KH, KW, pad, Ntot;
    const float* xb;
    const float* tile_ptr;   // channel ci0 of the (clamped) input row, at column 2 ow
    int ih0, iw0, HW;
    int ci0, kh, kw, bk;
    int eoff;                // -1: the element left of the window exists, 0: the window starts at column 0
    int off;                 // kw - pad
    unsigned okmask;
    int pending_seek = 0;
    __device__ __forceinline__ void column4(int j) {
        const int jj = j < Ntot ? j : 0;
        const int opix = OH * OW;
        const int b = jj / opix, pix = jj - b * opix;
        const int oh = pix / OW, ow = pix - oh * OW;
        HW = H * W;
        xb = x + (long long)b * Cin * HW;
        ih0 = oh * 2 - pad;
        iw0 = ow * 2;
        eoff = iw0 > 0 ? -1 : 0;
        ci0 = -bk; kh = 0; kw = 0;
        if (pending_seek > 0) seek(pending_seek);
        tile_ptr = xb + iw0; off = 0; okmask = 0;
    }
    __device__ __forceinline__ void seek(int t0) {
        const int k0 = t0 * bk, tap = k0 / Cin;
        ci0 = k0 - tap * Cin - bk;
        kh = tap / KW; kw = tap - kh * KW;
    }
    __device__ __forceinline__ void begin_tile(int) {
        ci0 += bk;
        if (ci0 >= Cin) { ci0 = 0; if (++kw == KW) { kw = 0; ++kh; } }
        const int ih = ih0 + kh;
        const bool row_ok = (unsigned)ih < (unsigned)H;
        const int ihc = min(max(ih, 0), H - 1);
        tile_ptr = xb + (ci0 * H + ihc) * W + iw0;
        off = kw - pad;
        okmask = !row_ok ? 0u : ((off < 0 && iw0 == 0) ? 0xeu : 0xfu);
    }
    __device__ __forceinline__ Raw load4(int k) const {
        const float* r = tile_ptr + (k & (bk - 1)) * HW;
        Raw q;
        q.a = *reinterpret_cast<const float4*>(r);
        q.b = *reinterpret_cast<const float4*>(r + 4);
        q.c = make_float4(r[eoff], 0.0f, 0.0f, 0.0f);
        return q;
    }
    struct Info { int off; unsigned okmask; };
    __device__ __forceinline__ Info info() const { return Info{off, okmask}; }
    __device__ __forceinline__ float4 pick(const Raw& q, int o, unsigned ok) const {
        // The nine staged values as opaque REGISTER values before the selects: hipcc otherwise rewrites "select of two fields" as "load from a
        // selected address", which keeps the staged rows in scratch instead of registers (560 B per lane, the kernel twice as slow).
        float e = q.c.x, a0 = q.a.x, a1 = q.a.y, a2 = q.a.z, a3 = q.a.w, b0 = q.b.x, b1 = q.b.y, b2 = q.b.z, b3 = q.b.w;
        asm volatile("" : "+v"(e), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
        float4 v;
        v.x = o < 0 ? e : (o > 0 ? a1 : a0);
        v.y = o < 0 ? a1 : (o > 0 ? a3 : a2);
        v.z = o < 0 ? a3 : (o > 0 ? b1 : b0);
        v.w = o < 0 ? b1 : (o > 0 ? b3 : b2);
        v.x = (ok & 1u) ? v.x : 0.0f; v.y = (ok & 2u) ? v.y : 0.0f;
        v.z = (ok & 4u) ? v.z : 0.0f; v.w = (ok & 8u) ? v.w : 0.0f;
        return v;
    }
    __device__ __forceinline__ float4 finish(const Raw& q, int, const Info& in) const { return pick(q, in.off, in.okmask); }
    __device__ __forceinline__ float4 finish(const Raw& q, int) const { return pick(q, off, okmask); }
};

// Vector stager for weights in their own (ci,kh,kw) order -- the 7x7/2 stem (Cin = 3: no tap holds a whole K-step).  Each
// staged row decodes its own (ci,kh,kw) (divisions by compile-time constants), loads 4 output pixels of one output row as
// four clamped scalars (unconditional), and remembers the in-image mask of the pass for fix().
template <int KH_, int KW_, int STRIDE>
struct LoaderIm2colRow4 {
    const float* x;
    int Cin, H, W, OH, OW, pad, K, Ntot;
    const float* xb;
    int ih0, iw0, HW;
    unsigned okmask[4];
    int pass;
    __device__ __forceinline__ void column4(int j) {
        const int jj = j < Ntot ? j : 0;
        const int opix = OH * OW;
        const int b = jj / opix, pix = jj - b * opix;
        const int oh = pix / OW, ow = pix - oh * OW;
        HW = H * W;
        xb = x + (long long)b * Cin * HW;
        ih0 = oh * STRIDE - pad;
        iw0 = ow * STRIDE - pad;
        pass = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) okmask[p] = 0;
    }
    __device__ __forceinline__ void begin_tile(int) { pass = 0; }
    __device__ __forceinline__ float4 load4(int k) {
        const int kc = min(k, K - 1);                 // rows k >= K meet zero weights
        const int ci = kc / (KH_ * KW_), rem = kc - ci * (KH_ * KW_);
        const int kh = rem / KW_, kw = rem - kh * KW_;
        const int ih = ih0 + kh, iw = iw0 + kw;
        const bool row_ok = (unsigned)ih < (unsigned)H;
        const float* r = xb + (ci * H + min(max(ih, 0), H - 1)) * W;
        const int c0 = min(max(iw, 0), W - 1), c1 = min(max(iw + STRIDE, 0), W - 1);
        const int c2 = min(max(iw + 2 * STRIDE, 0), W - 1), c3 = min(max(iw + 3 * STRIDE, 0), W - 1);
        const unsigned m = !row_ok ? 0u : ((unsigned)iw < (unsigned)W ? 1u : 0u) | ((unsigned)(iw + STRIDE) < (unsigned)W ? 2u : 0u) |
                                          ((unsigned)(iw + 2 * STRIDE) < (unsigned)W ? 4u : 0u) | ((unsigned)(iw + 3 * STRIDE) < (unsigned)W ? 8u : 0u);
#pragma unroll
        for (int p = 0; p < 4; ++p) if (p == pass) okmask[p] = m;      // pass is a compile-time constant after unrolling
        ++pass;
        return make_float4(r[c0], r[c1], r[c2], r[c3]);
    }
    __device__ __forceinline__ void fix(float4& v, int p) const {
        const unsigned m = okmask[p & 3];
        v.x = (m & 1u) ? v.x : 0.0f; v.y = (m & 2u) ? v.y : 0.0f; v.z = (m & 4u) ? v.z : 0.0f; v.w = (m & 8u) ? v.w : 0.0f;
    }
};

// BN(eval) + residual + ReLU.  Per-row operands and the residual are fetched first (clamped addresses, independent
// loads), arithmetic and stores follow: no load -> wait -> store chain per accumulator register.
struct EpiConv {
    const float* scale;
    const float* shift;
    const float* residual;
    float* y;
    int Cout, opix, Ntot, relu;
    __device__ __forceinline__ void tile(int mrow0, int j, const f32x16& acc) {
        const bool col_ok = j < Ntot;
        const int jj = col_ok ? j : 0;
        const int b = jj / opix, pix = jj - b * opix;
        const long long o0 = (long long)b * Cout * opix + pix;
        float sc[16], sh[16], v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mc = min(mrow0 + (r & 3) + 8 * (r >> 2), Cout - 1);
            sc[r] = scale[mc];
            sh[r] = shift[mc];
        }
        if (residual) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = residual[o0 + (long long)min(mrow0 + (r & 3) + 8 * (r >> 2), Cout - 1) * opix];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += acc[r] * sc[r] + sh[r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[r] * sc[r] + sh[r];
        }
        if (relu) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.0f);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + (r & 3) + 8 * (r >> 2);
            if (col_ok && m < Cout) y[o0 + (long long)m * opix] = v[r];
        }
    }
};

template <class Cfg, bool TAP>
__global__ __launch_bounds__(Cfg::THREADS) void conv2d_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ residual, float* __restrict__ y, int Cin,
                                                               int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride,
                                                               int pad, int Ntot, int relu) {
    extern __shared__ float lds[];
    const int K = Cin * KH * KW;
    LoaderWtC la{Wt, K, Cout};
    EpiConv ep{scale, shift, residual, y, Cout, OH * OW, Ntot, relu};
    if (TAP) {
        LoaderIm2colTap lb;
        lb.x = x; lb.Cin = Cin; lb.H = H; lb.W = W; lb.OH = OH; lb.OW = OW; lb.KH = KH; lb.KW = KW; lb.stride = stride;
        lb.pad = pad; lb.K = K; lb.Ntot = Ntot; lb.bk = Cfg::BK;
        mfma_gemm_block<Cfg>(lds, la, lb, ep, K, blockIdx.y * Cfg::BM, blockIdx.x * Cfg::BN);
    } else {
        LoaderIm2col lb{x, Cin, H, W, OH, OW, KH, KW, stride, pad, K, Ntot, nullptr, 0, 0, false};
        mfma_gemm_block<Cfg>(lds, la, lb, ep, K, blockIdx.y * Cfg::BM, blockIdx.x * Cfg::BN);
    }
}

// XCD-aware tile order.  The dispatcher hands consecutive workgroups to the 8 XCDs round-robin, each with its own L2.
// Logical tile lid = (id % 8) * (L / 8) + id / 8 gives every XCD a CONTIGUOUS range of tiles, walked with the Cout tiles
// fastest: the workgroups that share an input tile (other output channels) and the neighbouring output rows (whose 3x3
// windows overlap by two input rows) run on the same XCD close in time, so the im2col re-reads hit that XCD's L2.
// Measured with FETCH_SIZE on the ResNet-34 pass: 1767 -> 1056 MiB per pass (stem 58.7 -> 15.2, stages 1-2 64.7 -> 23.7 per
// launch).  When the layer's weights alone exceed an L2 (stage 4: 9.4 MB) neither order keeps them resident and the plain
// dispatch order measured best (39.6 MiB vs 47.7 / 107.7 per launch), so it is kept there.
__device__ __forceinline__ void conv_tile(int& tx, int& ty, long long weight_bytes) {
    const int gx = gridDim.x, gy = gridDim.y, L = gx * gy;
    const int id = blockIdx.x + blockIdx.y * gx;
    if (weight_bytes > (3ll << 20) || L % 8 != 0) { tx = blockIdx.x; ty = blockIdx.y; return; }
    const int lid = (id & 7) * (L >> 3) + (id >> 3);
    ty = lid % gy;
    tx = lid / gy;
}

// split-K: raw partial sums of K-slice z, same [b][Cout][pix] layout as y
struct EpiPartial {
    float* part;
    int Cout, opix, Ntot;
    __device__ __forceinline__ void tile(int mrow0, int j, const f32x16& acc) {
        if (j >= Ntot) return;
        const int b = j / opix, pix = j - b * opix;
        float* o = part + (long long)b * Cout * opix + pix;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + (r & 3) + 8 * (r >> 2);
            if (m < Cout) o[(long long)m * opix] = acc[r];
        }
    }
};

// y = relu?( scale * (part_0 + part_1 + ...) + shift + residual ), partials added in slice order (deterministic)
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const float* __restrict__ part, int splits, long long slice,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ residual, float* __restrict__ y, int Cout,
                                                                 int opix, int relu) {
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;      // opix % 4 == 0: one channel per float4
    if (i4 >= slice) return;
    float4 a = *reinterpret_cast<const float4*>(part + i4);
    for (int z = 1; z < splits; ++z) {
        const float4 p = *reinterpret_cast<const float4*>(part + z * slice + i4);
        a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
    }
    const int m = (int)((i4 / opix) % Cout);
    const float sc = scale[m], sh = shift[m];
    float4 v = make_float4(a.x * sc + sh, a.y * sc + sh, a.z * sc + sh, a.w * sc + sh);
    if (residual) { const float4 r = *reinterpret_cast<const float4*>(residual + i4); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<float4*>(y + i4) = v;
}

template <class Cfg, int STRIDE, int DEPTH = 1>
__global__ __launch_bounds__(Cfg::THREADS) void conv2d_vec_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   const float* __restrict__ residual, float* __restrict__ y, int Cin,
                                                                   int H, int W, int Cout, int OH, int OW, int KH, int KW, int pad,
                                                                   int Ntot, int relu, float* __restrict__ part, int splits) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = Cin * KH * KW;
    LoaderWt4 la{Wt, K, Cout};
    std::conditional_t<STRIDE == 22, LoaderIm2colTap4W2, LoaderIm2colTap4<STRIDE == 22 ? 2 : STRIDE>> lb;      // 22: stride 2, window loads
    lb.x = x; lb.Cin = Cin; lb.H = H; lb.W = W; lb.OH = OH; lb.OW = OW; lb.KH = KH; lb.KW = KW;
    lb.pad = pad; lb.Ntot = Ntot; lb.bk = Cfg::BK;
    int tx, ty;
    conv_tile(tx, ty, (long long)K * Cout * 4);
    if (splits <= 1) {
        EpiConv ep{scale, shift, residual, y, Cout, OH * OW, Ntot, relu};
        if (DEPTH == 2) mfma_gemm_block_vec2<Cfg>(lds, la, lb, ep, K, ty * Cfg::BM, tx * Cfg::BN);
        else mfma_gemm_block_vec<Cfg>(lds, la, lb, ep, K, ty * Cfg::BM, tx * Cfg::BN);
    } else {      // K-slice blockIdx.z: raw partial sums, combined by conv_splitk_reduce_kernel
        const int T = K / Cfg::BK, z = blockIdx.z;
        const int t0 = (int)((long long)T * z / splits), t1 = (int)((long long)T * (z + 1) / splits);
        EpiPartial ep{part + (long long)z * Cout * Ntot, Cout, OH * OW, Ntot};
        lb.pending_seek = t0;
        if (DEPTH == 2) mfma_gemm_block_vec2<Cfg>(lds, la, lb, ep, K, ty * Cfg::BM, tx * Cfg::BN, t0, t1);
        else mfma_gemm_block_vec<Cfg>(lds, la, lb, ep, K, ty * Cfg::BM, tx * Cfg::BN, t0, t1);
    }
}

// the 7x7 stride-2 pad-3 stem on the vector stager
template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void conv2d_stem_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    const float* __restrict__ residual, float* __restrict__ y, int Cin,
                                                                    int H, int W, int Cout, int OH, int OW, int pad, int Ntot, int relu) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = Cin * 49;
    LoaderWt4 la{Wt, K, Cout};
    EpiConv ep{scale, shift, residual, y, Cout, OH * OW, Ntot, relu};
    LoaderIm2colRow4<7, 7, 2> lb;
    lb.x = x; lb.Cin = Cin; lb.H = H; lb.W = W; lb.OH = OH; lb.OW = OW; lb.pad = pad; lb.K = K; lb.Ntot = Ntot;
    int tx, ty;
    conv_tile(tx, ty, (long long)K * Cout * 4);
    mfma_gemm_block_vec<Cfg>(lds, la, lb, ep, K, ty * Cfg::BM, tx * Cfg::BN);
}

template <class Cfg>
void launch_conv_vec(const float* x, const float* Wt, const float* scale, const float* shift, const float* residual, float* y,
                     int Cin, int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride, int pad, int Ntot, int relu,
                     float* part, int splits, hipStream_t st) {
    const dim3 grid(di2p_cdiv(Ntot, Cfg::BN), di2p_cdiv(Cout, Cfg::BM), splits), block(Cfg::THREADS);
    const size_t lds = Cfg::LDS_FLOATS * sizeof(float);
    // depth-2 register prefetch (bit-identical results; +1..10 % per layer, most on the stride-2 layers); conv_depth1 = 1 selects
    // the depth-1 engine (tests compare the two)
    const bool depth1 = di2p_opt(DI2P_OPT_CONV_DEPTH1) != 0;
#define DI2P_CONV_VEC_LAUNCH(S, D) hipLaunchKernelGGL((conv2d_vec_kernel<Cfg, S, D>), grid, block, lds, st, x, Wt, scale, shift, residual, y, Cin, H, W, Cout, OH, OW, KH, KW, pad, Ntot, relu, part, splits)
    // stride 2: aligned 8-float windows where the shape allows (W == 2 OW, whole 8-column groups, padded 3x3 or plain 1x1); conv_s2scalar = 1: never
    const bool window = stride == 2 && W == 2 * OW && W % 8 == 0 && ((pad == 1 && KW == 3) || (pad == 0 && KW == 1)) &&
                        ((uintptr_t)x & 15) == 0 && !di2p_opt(DI2P_OPT_CONV_S2SCALAR);
    if (stride == 1) { if (depth1) DI2P_CONV_VEC_LAUNCH(1, 1); else DI2P_CONV_VEC_LAUNCH(1, 2); }
    else if (window) { if (depth1) DI2P_CONV_VEC_LAUNCH(22, 1); else DI2P_CONV_VEC_LAUNCH(22, 2); }
    else { if (depth1) DI2P_CONV_VEC_LAUNCH(2, 1); else DI2P_CONV_VEC_LAUNCH(2, 2); }
#undef DI2P_CONV_VEC_LAUNCH
    if (splits > 1) {
        const long long slice = (long long)Cout * Ntot;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((slice / 4 + 255) / 256)), dim3(256), 0, st, part, splits, slice,
                           scale, shift, residual, y, Cout, OH * OW, relu);
    }
}

template <class Cfg>
void launch_conv(const float* x, const float* Wt, const float* scale, const float* shift, const float* residual, float* y,
                 int Cin, int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride, int pad, int Ntot, int relu,
                 int tap_major, hipStream_t st) {
    const dim3 grid(di2p_cdiv(Ntot, Cfg::BN), di2p_cdiv(Cout, Cfg::BM)), block(Cfg::THREADS);
    const size_t lds = Cfg::LDS_FLOATS * sizeof(float);
    if (tap_major)
        hipLaunchKernelGGL((conv2d_kernel<Cfg, true>), grid, block, lds, st, x, Wt, scale, shift, residual, y, Cin, H, W, Cout,
                           OH, OW, KH, KW, stride, pad, Ntot, relu);
    else
        hipLaunchKernelGGL((conv2d_kernel<Cfg, false>), grid, block, lds, st, x, Wt, scale, shift, residual, y, Cin, H, W, Cout,
                           OH, OW, KH, KW, stride, pad, Ntot, relu);
}

__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int OH,
                                                           int OW, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ow = (int)(i % OW);
    const long long t = i / OW;
    const int oh = (int)(t % OH);
    const long long bc = t / OH;
    const float* p = x + bc * H * W;
    float m = -__builtin_inff();
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
        const int ih = oh * 2 - 1 + dh;
        if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
            const int iw = ow * 2 - 1 + dw;
            if ((unsigned)iw < (unsigned)W) m = fmaxf(m, p[(long long)ih * W + iw]);
        }
    }
    y[i] = m;
}

// 4 consecutive outputs per lane (W % 8 == 0, 16-byte aligned rows): per input row two aligned 16-byte loads + the one
// element to the left instead of 9 scalar loads per output; one 16-byte store.  Same max tree as the scalar kernel.
__global__ __launch_bounds__(256) void maxpool3x3s2_vec4_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                                int OH, int OW, long long total4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int ow4 = OW >> 2;
    const int og = (int)(i % ow4);
    const long long t = i / ow4;
    const int oh = (int)(t % OH);
    const long long bc = t / OH;
    const float* p = x + bc * H * W;
    const int iw0 = og * 8;                       // input column of output 4*og, kernel column 1
    const float ninf = -__builtin_inff();
    float m0 = ninf, m1 = ninf, m2 = ninf, m3 = ninf;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
        const int ih = oh * 2 - 1 + dh;
        if ((unsigned)ih >= (unsigned)H) continue;
        const float* r = p + (long long)ih * W + iw0;
        const float4 a = *reinterpret_cast<const float4*>(r);
        const float4 b = *reinterpret_cast<const float4*>(r + 4);
        const float l = iw0 > 0 ? r[-1] : ninf;
        // output j covers input columns iw0 + 2j - 1 .. iw0 + 2j + 1, visited left to right like the scalar kernel
        m0 = fmaxf(fmaxf(fmaxf(m0, l), a.x), a.y);
        m1 = fmaxf(fmaxf(fmaxf(m1, a.y), a.z), a.w);
        m2 = fmaxf(fmaxf(fmaxf(m2, a.w), b.x), b.y);
        m3 = fmaxf(fmaxf(fmaxf(m3, b.y), b.z), b.w);
    }
    *reinterpret_cast<float4*>(y + (bc * OH + oh) * OW + og * 4) = make_float4(m0, m1, m2, m3);
}

// one wavefront per (b,c): mean over HW
__global__ __launch_bounds__(256) void global_avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int HW) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* r = x + row * HW;
    float s = 0.0f;
    for (int n = lane; n < HW; n += 64) s += r[n];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[row] = s / (float)HW;
}

}  // namespace

using CfgC128x128 = TileCfg<2, 2, 2, 2>;
using CfgC64x128 = TileCfg<2, 2, 1, 2>;
using CfgC64x64 = TileCfg<2, 2, 1, 1>;
using CfgC128x64 = TileCfg<2, 2, 2, 1>;
using CfgC128x128k32 = TileCfg<2, 2, 2, 2, 32>;
using CfgC64x128k32 = TileCfg<2, 2, 1, 2, 32>;
using CfgC64x64k32 = TileCfg<2, 2, 1, 1, 32>;
using CfgC128x64k32 = TileCfg<2, 2, 2, 1, 32>;

namespace {
// split-K factor the dispatcher uses for a shape (1 = none): the vector path with 64x64 tiles leaves the small-N layers
// (ResNet stage 4 at 160x512: 320 workgroups for 256 CUs) latency-bound; K-slices along the filter taps give the chip
// >= 3 workgroups per CU.  Partials are combined in slice order by a separate pass (deterministic, no atomics).
int conv_splits(int Cin, int Cout, int KH, int KW, int OW, long long opix, int tap_major) {
    if (di2p_opt(DI2P_OPT_CONV_NOSPLIT) || !tap_major || Cin % 32 != 0 || OW % 4 != 0 || Cout % 4 != 0 || Cout < 128) return 1;
    // The decision depends on the PER-FRAME shape only, never on the batch size: a frame's result must not change with
    // the batch it is computed in (data-parallel shards reproduce the unsharded run bit for bit).  The limit is the
    // number of 64x64 workgroups one frame contributes below which a 32-frame batch leaves CUs idle.
    const long long per_frame = di2p_cdiv(opix, 64) * (long long)di2p_cdiv(Cout, 64);
    const int T = Cin * KH * KW / 32;
    const long long limit = di2p_opt(DI2P_OPT_CONV_SPLIT_BLOCKS);   // tuning knob (per-frame workgroups), default 32
    if (per_frame >= limit || T < 24) return 1;
    return per_frame * 2 >= limit * 3 / 2 ? 2 : 3;
}

int conv2d_impl(const float* x, const float* Wt, const float* scale, const float* shift, const float* residual,
                float* y, int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad, int relu,
                int tap_major, void* workspace, long long workspace_bytes, void* stream) {
    DI2P_CHECK_ARG(x && Wt && scale && shift && y, "null pointer");
    DI2P_CHECK_ARG(B >= 0 && Cin >= 1 && H >= 1 && W >= 1 && Cout >= 1 && KH >= 1 && KW >= 1 && stride >= 1 && pad >= 0, "bad size");
    if (B == 0) return 0;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    DI2P_CHECK_ARG(OH >= 1 && OW >= 1, "empty output");
    DI2P_CHECK_ARG(!tap_major || Cin % 16 == 0, "tap-major weights need Cin % 16 == 0");
    DI2P_CHECK_ARG((long long)Cin * H * W < (1ll << 31), "per-image extent must fit 31 bits");
    const long long Ntot_ll = (long long)B * OH * OW;
    DI2P_CHECK_ARG(Ntot_ll < (1ll << 31), "too many output pixels");
    const int Ntot = (int)Ntot_ll;
    hipStream_t st = (hipStream_t)stream;
#define DI2P_CONV(CFG) launch_conv<CFG>(x, Wt, scale, shift, residual, y, Cin, H, W, Cout, OH, OW, KH, KW, stride, pad, Ntot, relu, tap_major, st)
    // tile choice: K-step 32 when a tap holds whole 32-channel groups (halves barriers, doubles the prefetch
    // distance); the largest tile that still yields >= ~2 workgroups per CU.  DI2P_CONV_CFG overrides (experiments).
    const int force = (int)di2p_opt(DI2P_OPT_CONV_CFG);
    const bool k32 = tap_major && Cin % 32 == 0;
    const long long nb64x128 = (long long)di2p_cdiv(Ntot, 128) * di2p_cdiv(Cout, 64);
    const long long nb128x128 = (long long)di2p_cdiv(Ntot, 128) * di2p_cdiv(Cout, 128);
    const long long nb128x64 = (long long)di2p_cdiv(Ntot, 64) * di2p_cdiv(Cout, 128);
    int choice;   // 0: 64x64  1: 64x128  2: 128x64  3: 128x128
    if (Cout >= 128 && nb128x128 >= 512) choice = 3;
    else if (nb64x128 >= 512) choice = 1;
    else if (Cout >= 128 && nb128x64 >= 512) choice = 2;
    else choice = 0;
    if (force >= 0) { choice = force % 10; if (choice >= 2 && Cout < 128) choice = 1; }
    const bool use32 = k32 && (force < 0 || force >= 10);
    // the 7x7/2 stem (weights in their own order): row-decoding vector stager
    if (!tap_major && KH == 7 && KW == 7 && stride == 2 && OW % 4 == 0 && Cout % 4 == 0 && ((uintptr_t)Wt & 15) == 0 &&
        !di2p_opt(DI2P_OPT_CONV_NOVEC)) {
        using CfgS = TileCfg<2, 2, 1, 2, 32>;      // 64 x 128, B_PASSES = 4 (one mask per pass)
        const dim3 grid(di2p_cdiv(Ntot, CfgS::BN), di2p_cdiv(Cout, CfgS::BM));
        hipLaunchKernelGGL(conv2d_stem_kernel<CfgS>, grid, dim3(CfgS::THREADS), CfgS::LDS_FLOATS * sizeof(float), st, x, Wt, scale, shift,
                           residual, y, Cin, H, W, Cout, OH, OW, pad, Ntot, relu);
        DI2P_RETURN_LAUNCH();
    }
    // vector stager: tap-major weights, 32-channel taps, whole 4-pixel groups per output row, 16-byte aligned weights
    const bool novec = di2p_opt(DI2P_OPT_CONV_NOVEC) != 0;
    const bool vec = !novec && use32 && OW % 4 == 0 && Cout % 4 == 0 && ((uintptr_t)Wt & 15) == 0 &&
                     ((stride == 1 && pad <= 1 && KW <= 2 * pad + 1 && W >= 4) || stride == 2);
    if (vec && force < 0) choice = Cout <= 64 ? 1 : 0;   // measured: 64x64 tiles (more, smaller workgroups) win for Cout >= 128
    int splits = vec ? conv_splits(Cin, Cout, KH, KW, OW, (long long)OH * OW, tap_major) : 1;
    if (splits > 1 && (!workspace || workspace_bytes < (long long)splits * Cout * Ntot * (long long)sizeof(float) || ((uintptr_t)workspace & 15))) splits = 1;
    if (splits > 1) choice = 0;
    float* part = (float*)workspace;
#define DI2P_CONVV(CFG) launch_conv_vec<CFG>(x, Wt, scale, shift, residual, y, Cin, H, W, Cout, OH, OW, KH, KW, stride, pad, Ntot, relu, part, splits, st)
    if (vec) {
        switch (choice) { case 3: DI2P_CONVV(CfgC128x128k32); break; case 2: DI2P_CONVV(CfgC128x64k32); break;
                          case 1: DI2P_CONVV(CfgC64x128k32); break; default: DI2P_CONVV(CfgC64x64k32); }
    } else if (use32) {
        switch (choice) { case 3: DI2P_CONV(CfgC128x128k32); break; case 2: DI2P_CONV(CfgC128x64k32); break;
                          case 1: DI2P_CONV(CfgC64x128k32); break; default: DI2P_CONV(CfgC64x64k32); }
    } else {
        switch (choice) { case 3: DI2P_CONV(CfgC128x128); break; case 2: DI2P_CONV(CfgC128x64); break;
                          case 1: DI2P_CONV(CfgC64x128); break; default: DI2P_CONV(CfgC64x64); }
    }
#undef DI2P_CONV
#undef DI2P_CONVV
    DI2P_RETURN_LAUNCH();
}
}  // namespace

extern "C" int di2p_conv2d(const float* x, const float* Wt, const float* scale, const float* shift, const float* residual,
                           float* y, int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad, int relu,
                           int tap_major, void* stream) {
    return conv2d_impl(x, Wt, scale, shift, residual, y, B, Cin, H, W, Cout, KH, KW, stride, pad, relu, tap_major, nullptr, 0, stream);
}

extern "C" int di2p_conv2d_ws(const float* x, const float* Wt, const float* scale, const float* shift, const float* residual,
                              float* y, int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad, int relu,
                              int tap_major, void* workspace, long long workspace_bytes, void* stream) {
    return conv2d_impl(x, Wt, scale, shift, residual, y, B, Cin, H, W, Cout, KH, KW, stride, pad, relu, tap_major, workspace,
                       workspace_bytes, stream);
}

extern "C" long long di2p_conv2d_workspace_bytes(int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                                                 int tap_major) {
    if (B <= 0 || Cin < 1 || H < 1 || W < 1 || Cout < 1 || KH < 1 || KW < 1 || stride < 1 || pad < 0) return 0;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    if (OH < 1 || OW < 1) return 0;
    const bool shape_ok = (stride == 1 && pad <= 1 && KW <= 2 * pad + 1 && W >= 4) || stride == 2;
    const long long Ntot = (long long)B * OH * OW;
    const int splits = shape_ok ? conv_splits(Cin, Cout, KH, KW, OW, (long long)OH * OW, tap_major) : 1;
    return splits > 1 ? (long long)splits * Cout * Ntot * (long long)sizeof(float) : 0;
}

extern "C" int di2p_maxpool3x3s2(const float* x, float* y, int B, int C, int H, int W, void* stream) {
    DI2P_CHECK_ARG(x && y && B >= 0 && C >= 1 && H >= 1 && W >= 1, "bad args");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)B * C * OH * OW;
    if (total == 0) return 0;
    if (W % 8 == 0 && OW * 2 == W && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        hipLaunchKernelGGL(maxpool3x3s2_vec4_kernel, dim3(di2p_cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, OH, OW, total / 4);
        DI2P_RETURN_LAUNCH();
    }
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(di2p_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, OH, OW, total);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_global_avgpool(const float* x, float* y, int B, int C, int HW, void* stream) {
    DI2P_CHECK_ARG(x && y && B >= 0 && C >= 1 && HW >= 1, "bad args");
    const long long rows = (long long)B * C;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(global_avgpool_kernel, dim3(di2p_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, rows, HW);
    DI2P_RETURN_LAUNCH();
}
