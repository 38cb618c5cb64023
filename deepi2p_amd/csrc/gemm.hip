// Pointwise contraction kernels (EquivariantLayer / MyConv2d 1x1 + BN(eval) + ReLU) on fp32 MFMA.
//
// Replaces the cuDNN/ATen conv1d/conv2d(1x1)+batch_norm+relu chains of models/layers_pc.py:259-342,
// :110-190 and the torch.cat / expand / gather tensors that feed them (models/networks_pc.py:98,
// models/layers_pc.py:808-813, models/networks_united.py:139-197): concatenation, gather-by-index
// and group-broadcast are done by the B-operand loader, BN/bias/ReLU/max-over-K/interpolated-add by
// the epilogue, so none of those intermediates ever exists in HBM.
#include "mfma_tile.h"

#include <stdint.h>
#include <stdlib.h>

namespace {

struct LoaderWt {  // packed weight [K][M]
    const float* Wt;
    int K, M;
    __device__ __forceinline__ float load(int k, int m) const { return (k < K && m < M) ? Wt[k * M + m] : 0.0f; }   // K*M < 2^31 (host-checked)
};

struct SrcDev {
    const float* ptr[DI2P_MAX_SRC];
    const int* gidx[DI2P_MAX_SRC];
    long long batch_stride[DI2P_MAX_SRC];
    int row_stride[DI2P_MAX_SRC];
    int c_end[DI2P_MAX_SRC];  // exclusive prefix end of each source's channel range
    int mode[DI2P_MAX_SRC];
    int group[DI2P_MAX_SRC];
    int n_src;
};

struct LoaderConcat {
    SrcDev s;
    int b, N, K;
    bool valid;
    const float* base[DI2P_MAX_SRC];  // per-column base pointer (already offset by batch and column)
    __device__ __forceinline__ void column(int n) {
        valid = n < N;
#pragma unroll
        for (int i = 0; i < DI2P_MAX_SRC; ++i) {
            base[i] = nullptr;
            if (i < s.n_src && valid) {
                int off = n;
                if (s.mode[i] == DI2P_SRC_GATHER) off = s.gidx[i][(long long)b * N + n];
                else if (s.mode[i] == DI2P_SRC_GROUP) off = n / s.group[i];
                base[i] = s.ptr[i] + (long long)b * s.batch_stride[i] + off;
            }
        }
    }
    __device__ __forceinline__ void begin_tile(int) {}
    __device__ __forceinline__ float load(int k) const {
        k = __builtin_amdgcn_readfirstlane(k);  // a wave stages one panel row: k is wave-uniform
        if (!valid || k >= K) return 0.0f;
        // within-frame offsets fit 32 bits (checked on the host): no 64-bit multiplies in the hot loader
        if (k < s.c_end[0]) return base[0][k * s.row_stride[0]];
        if (DI2P_MAX_SRC > 1 && k < s.c_end[1]) return base[1][(k - s.c_end[0]) * s.row_stride[1]];
        return base[2][(k - s.c_end[1]) * s.row_stride[2]];
    }
};

// 4-column stager.  Loads are UNCONDITIONAL from clamped addresses (no control flow in the K-loop): rows k >= K meet
// zero weights (clamping k re-reads real, column-local data), columns n >= N are dropped by the epilogue.  The source
// of a row is picked with selects.  DENSE: every source is DENSE and 16-byte addressable -> one 16-byte load per row;
// otherwise four dword loads at per-source column offsets (dense n, gathered gidx[n], group n/group).
template <bool DENSE>
struct LoaderConcat4 {
    // Per-source state as NAMED SCALARS: with arrays (base[3], rs[3], off[3][4]) the source select of load4() -- s2 ? base[2] : ... --
    // was canonicalised by hipcc into a load from a dynamically indexed private array, i.e. the whole loader lived in scratch
    // (208-240 B per lane, 16 scratch loads per K-step) although nothing was ever spilled.
    // Source 0's values plus the INCREMENTS to source 1 and from source 1 to source 2: load4() picks the source of a row as
    // v0 + (s1 ? d1 : 0) + (s2 ? d2 : 0).  A select whose arms are two loaded members (s2 ? b2 : b1) is rewritten by hipcc into a
    // load from a select of the members' addresses -- a dynamic index that keeps the loader object in private memory.
    const float* b0;
    long long db1, db2;          // byte increments of the base pointer
    int r0, dr1, dr2;            // row strides
    int o0[4], do1[4], do2[4];   // !DENSE: column offsets of the lane's 4 columns (static indices only)
    int c0, c1, K;               // channel range ends of source 0 / 1 (== K when absent)
    SrcDev s;
    int b, N;
    __device__ __forceinline__ void one_source(int i, int nc, const float*& base, int& rs, int* q) const {
        // i is a compile-time constant at every call site.  Absent sources were filled in on the host as aliases of source 0 with an
        // empty channel range (alias_absent_sources), so they are valid to address and never selected.
        base = s.ptr[i] + (long long)b * s.batch_stride[i] + (DENSE ? nc : 0);
        rs = s.row_stride[i];
        if (!DENSE) {
            const int mode = s.mode[i];
            const int* gi = s.gidx[i];
            const int grp = s.group[i];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                int o = nc + c;
                if (mode == DI2P_SRC_GATHER) o = gi[(long long)b * N + nc + c];
                else if (mode == DI2P_SRC_GROUP) o = (nc + c) / grp;
                q[c] = o;
            }
        }
    }
    __device__ __forceinline__ void column4(int n) {
        const int nc = min(n, N - 4);
        const float *p0, *p1, *p2;
        int s0, s1, s2, q0[4] = {0, 0, 0, 0}, q1[4] = {0, 0, 0, 0}, q2[4] = {0, 0, 0, 0};
        one_source(0, nc, p0, s0, q0);
        one_source(1, nc, p1, s1, q1);
        one_source(2, nc, p2, s2, q2);
        b0 = p0;
        db1 = (long long)(reinterpret_cast<uintptr_t>(p1) - reinterpret_cast<uintptr_t>(p0));
        db2 = (long long)(reinterpret_cast<uintptr_t>(p2) - reinterpret_cast<uintptr_t>(p1));
        r0 = s0; dr1 = s1 - s0; dr2 = s2 - s1;
#pragma unroll
        for (int c = 0; c < 4; ++c) { o0[c] = q0[c]; do1[c] = q1[c] - q0[c]; do2[c] = q2[c] - q1[c]; }
        c0 = s.c_end[0];
        c1 = s.c_end[1];        // == K when source 1 is absent (prefix ends, filled on the host)
    }
    __device__ __forceinline__ void begin_tile(int) {}
    __device__ __forceinline__ float4 load4(int k) const {
        const int kc = min(k, K - 1);
        const bool s1 = kc >= c0, s2 = kc >= c1;          // s2 implies s1 (c1 >= c0)
        const int kk = kc - (s1 ? c0 : 0) - (s2 ? c1 - c0 : 0);
        const int r = r0 + (s1 ? dr1 : 0) + (s2 ? dr2 : 0);
        const long long byte_off = (s1 ? db1 : 0ll) + (s2 ? db2 : 0ll) + (long long)(kk * r) * 4;      // kk*r < 2^31 (host-checked)
        const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(b0) + byte_off);
        if (DENSE) return *reinterpret_cast<const float4*>(p);
        int q[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] = o0[c] + (s1 ? do1[c] : 0) + (s2 ? do2[c] : 0);
        return make_float4(p[q[0]], p[q[1]], p[q[2]], p[q[3]]);
    }
    __device__ __forceinline__ void fix(float4&, int) const {}
    struct Info {};
    __device__ __forceinline__ Info info() const { return Info{}; }
    __device__ __forceinline__ void fix(float4&, int, const Info&) const {}
};

// bf16x3 (see the section of that name below): the exact three-way split of fp32 values into truncated bf16 terms
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float x3_hi16(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u); }
// the bf16 (high halves) of two floats in one word: low half <- x0, high half <- x1
__device__ __forceinline__ unsigned x3_pack_hi(float x0, float x1) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
}
// four consecutive k of one column -> three planes of 4 x bf16
__device__ __forceinline__ void x3_split4(float f0, float f1, float f2, float f3, u32x2_t& p1, u32x2_t& p2, u32x2_t& p3) {
    const float r0 = f0 - x3_hi16(f0), r1 = f1 - x3_hi16(f1), r2 = f2 - x3_hi16(f2), r3 = f3 - x3_hi16(f3);
    const float q0 = r0 - x3_hi16(r0), q1 = r1 - x3_hi16(r1), q2 = r2 - x3_hi16(r2), q3 = r3 - x3_hi16(r3);
    p1 = u32x2_t{x3_pack_hi(f0, f1), x3_pack_hi(f2, f3)};
    p2 = u32x2_t{x3_pack_hi(r0, r1), x3_pack_hi(r2, r3)};
    p3 = u32x2_t{x3_pack_hi(q0, q1), x3_pack_hi(q2, q3)};
}

struct EpiDev {
    const float* scale;
    const float* shift;
    const float* batch_bias;
    const float* g_table[2];
    const int* g_idx[2];
    const float* g_w[2];
    int g_nodes[2];
    int g_k[2];      // neighbours per column of each gathered table (<= DI2P_MAX_GK)
    int relu;
    int group_max;
    int transpose_out;
    float* gmax_out;          // with group_max > 1: Y is stored in full AND the group maxima go here ([B,M,N/group_max])
    float* gmax_dst;          // where the group maxima go: gmax_out, or Y itself when only the maxima are stored (resolved on the host)
    u32x2_t* planes;          // the full-size output as three bf16 planes [B][3][M/4][N] x 4 consecutive rows (kernels instantiated with PLANES only)
};

// The accumulator tile of a lane is 16 rows of ONE column: rows mrow0 + 8g + {0..3}, g = 0..3.  All per-row operands
// (scale/shift/bias, the gathered tables stored [B][nodes][M]) are fetched as one float4 per group g from clamped
// addresses, before any arithmetic, so the epilogue is a handful of independent loads instead of 16 serial
// load -> wait -> store rounds.  Needs M % 4 == 0 whenever float4 operands are used (host-checked).
// GK0 / GK1 >= 0: the neighbour counts of the two gathered tables are compile-time constants AND M % 32 == 0 (the fused head: 3 + 3 on 128 rows);
// -1: run-time counts (any k, any M % 4 == 0).
template <int GK0 = -1, int GK1 = -1, bool PLANES = false>
struct EpiPointwiseT {
    EpiDev e;
    float* Y;
    int b, M, N;
    float* lds_tile = nullptr;   // fused chains: write the tile to LDS as the next layer's [k = m][n - lds_n0] operand panel
    int lds_n0 = 0, lds_ld = 0;
    // compile-time gathered tables: the neighbours of ONE column (row pointers at channel 0, weights), and their weighted sum on 16 rows
    static constexpr int NG = (GK0 > 0 ? GK0 : 0) + (GK1 > 0 ? GK1 : 0);
    struct Gathered {
        const float* gp[NG > 0 ? NG : 1];
        float gw[NG > 0 ? NG : 1];
    };
    __device__ __forceinline__ void gather_setup(int nc, Gathered& G) const {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int gk = t == 0 ? GK0 : GK1;
            if (gk <= 0) continue;
            const long long col = ((long long)b * N + nc) * gk;
            const float* gt = e.g_table[t] + (long long)b * e.g_nodes[t] * M;
#pragma unroll
            for (int j = 0; j < gk; ++j) {
                const int s = (t == 0 ? 0 : GK0) + j;
                G.gp[s] = gt + (long long)e.g_idx[t][col + j] * M;
                G.gw[s] = e.g_w[t] ? e.g_w[t][col + j] : 1.0f;
            }
        }
    }
    // One row group (4 rows from mrow) of every neighbour: request, and later reduce to t4 = sum_s gw[s] * G_s[mrow + {0..3}] -- an fma
    // chain from zero, tables in order, neighbours in order (the run-time path's chain).
    __device__ __forceinline__ void gather_issue(const Gathered& G, int mrow, float4 (&q)[NG > 0 ? NG : 1]) const {
#pragma unroll
        for (int s = 0; s < NG; ++s) q[s] = *reinterpret_cast<const float4*>(G.gp[s] + mrow);
    }
    __device__ __forceinline__ float4 gather_reduce(const Gathered& G, const float4 (&q)[NG > 0 ? NG : 1]) const {
        float4 t4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int s = 0; s < NG; ++s) {
            t4.x = fmaf(G.gw[s], q[s].x, t4.x); t4.y = fmaf(G.gw[s], q[s].y, t4.y);
            t4.z = fmaf(G.gw[s], q[s].z, t4.z); t4.w = fmaf(G.gw[s], q[s].w, t4.w);
        }
        return t4;
    }
    // t16 = the gathered sum on the 16 rows mrow0 + 8g + {0..3}
    __device__ __forceinline__ void gather_sum(const Gathered& G, int mrow0, float (&t16)[16]) const {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 q[NG > 0 ? NG : 1];
            gather_issue(G, mrow0 + 8 * g, q);
            const float4 t4 = gather_reduce(G, q);
            t16[4 * g + 0] = t4.x; t16[4 * g + 1] = t4.y; t16[4 * g + 2] = t4.z; t16[4 * g + 3] = t4.w;
        }
    }
    // the epilogue arithmetic proper (bias, gathered add, scale, shift, relu) on the 16 rows mrow0 + 8g + {0..3} of column nc (< N)
    __device__ __forceinline__ void apply(int mrow0, int nc, const f32x16& acc, float (&v)[16]) const {
        apply_pre(mrow0, nc, acc, nullptr, v);
    }
    // the per-row operand `row` (scale / shift / this frame's bias) on the 16 rows: ROWS4 (M % 4 == 0, the bf16x3 kernels) = one 16-byte load per
    // row group instead of four clamped dword loads (rows >= M, never stored, then see other rows' values: the stored ones are the same)
    template <bool ROWS4, class F>
    __device__ __forceinline__ void row_operand(const float* row, int mrow0, float (&v)[16], F f) const {
        if (ROWS4) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 q = *reinterpret_cast<const float4*>(row + min(mrow0 + 8 * g, M - 4));
                v[4 * g] = f(v[4 * g], q.x); v[4 * g + 1] = f(v[4 * g + 1], q.y); v[4 * g + 2] = f(v[4 * g + 2], q.z); v[4 * g + 3] = f(v[4 * g + 3], q.w);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = f(v[r], row[min(mrow0 + (r & 3) + 8 * (r >> 2), M - 1)]);
        }
    }
    // pre != nullptr (compile-time tables only): the gathered sum of these rows, already computed by gather_sum
    template <bool ROWS4 = false>
    __device__ __forceinline__ void apply_pre(int mrow0, int nc, const f32x16& acc, const float* pre, float (&v)[16]) const {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[r];
        if (e.batch_bias) row_operand<ROWS4>(e.batch_bias + (long long)b * M, mrow0, v, [](float a, float x) { return a + x; });
        // gathered add (per_point_pn layer 0): v[m] += sum_j w_j * G[b][idx_j][m]
        if (GK0 >= 0 && GK1 >= 0) {
            // The neighbours' products are summed FIRST, row group by row group, in fresh registers (t = sum_j w_j G_j: tables in order,
            // neighbours in order) and added to the accumulators ONCE.  With the accumulators as the running sum hipcc moved all sixteen of
            // them between the accumulation registers and the vector registers around every neighbour (32 moves per 16 fused multiply-adds)
            // and re-derived every 64-bit row address: ~1300 instructions per tile where ~200 do the work -- the fused head ran 12 vector
            // instructions per matrix instruction.  Row offsets are constants here (M % 32 == 0: every row group lies inside M).
            if (pre) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += pre[r];
            } else {
                constexpr int NG = GK0 + GK1;
                const float* gp[NG > 0 ? NG : 1];
                float gw[NG > 0 ? NG : 1];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int gk = t == 0 ? GK0 : GK1;
                    if (gk == 0) continue;
                    const long long col = ((long long)b * N + nc) * gk;
                    const float* gt = e.g_table[t] + (long long)b * e.g_nodes[t] * M + mrow0;
#pragma unroll
                    for (int j = 0; j < gk; ++j) {
                        const int s = (t == 0 ? 0 : GK0) + j;
                        gp[s] = gt + (long long)e.g_idx[t][col + j] * M;
                        gw[s] = e.g_w[t] ? e.g_w[t][col + j] : 1.0f;
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 q[NG > 0 ? NG : 1];
#pragma unroll
                    for (int s = 0; s < NG; ++s) q[s] = *reinterpret_cast<const float4*>(gp[s] + 8 * g);
                    float4 t4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);          // the same chain as the run-time path below: fma from zero, then one add
#pragma unroll
                    for (int s = 0; s < NG; ++s) {
                        t4.x = fmaf(gw[s], q[s].x, t4.x); t4.y = fmaf(gw[s], q[s].y, t4.y); t4.z = fmaf(gw[s], q[s].z, t4.z); t4.w = fmaf(gw[s], q[s].w, t4.w);
                    }
                    v[4 * g + 0] += t4.x; v[4 * g + 1] += t4.y; v[4 * g + 2] += t4.z; v[4 * g + 3] += t4.w;
                }
            }
        } else if (e.g_table[0] || e.g_table[1]) {
            float t16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t16[r] = 0.0f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (e.g_table[t]) {
                    const float* gt = e.g_table[t] + (long long)b * e.g_nodes[t] * M;
                    const int gk = e.g_k[t];
                    auto add_neighbour = [&](int j) {
                        const int gi = e.g_idx[t][((long long)b * N + nc) * gk + j];
                        const float gw = e.g_w[t] ? e.g_w[t][((long long)b * N + nc) * gk + j] : 1.0f;
                        const float* gp = gt + (long long)gi * M;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 q = *reinterpret_cast<const float4*>(gp + min(mrow0 + 8 * g, M - 4));
                            t16[4 * g + 0] = fmaf(gw, q.x, t16[4 * g + 0]); t16[4 * g + 1] = fmaf(gw, q.y, t16[4 * g + 1]);
                            t16[4 * g + 2] = fmaf(gw, q.z, t16[4 * g + 2]); t16[4 * g + 3] = fmaf(gw, q.w, t16[4 * g + 3]);
                        }
                    };
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < gk) add_neighbour(j);
                    for (int j = 4; j < gk; ++j) add_neighbour(j);       // k > 4 (the reference accepts any k): same order, rolled
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += t16[r];
        }
        if (e.scale) row_operand<ROWS4>(e.scale, mrow0, v, [](float a, float x) { return a * x; });
        if (e.shift) row_operand<ROWS4>(e.shift, mrow0, v, [](float a, float x) { return a + x; });
        if (e.relu) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.0f);
        }
    }
    // PLANES: the rows mrow0 + 8g + {0..3} of column n as three bf16 quads of 8 bytes each -- the k-quads of the layer that contracts over this
    // layer's rows (pointwise_gemm_x3p_kernel stages them without touching them); M % 4 == 0 (host-checked)
    __device__ __forceinline__ void store_planes(int mrow0, int n, const float (&v)[16]) const {
        const long long ps = (long long)(M >> 2) * N;               // one plane of one frame, in quads
        u32x2_t* p = e.planes + (long long)b * 3 * ps + n;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int m = mrow0 + 8 * g;
            if (m < M) {
                u32x2_t p1, p2, p3;
                x3_split4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3], p1, p2, p3);
                u32x2_t* q = p + (long long)(m >> 2) * N;
                q[0] = p1; q[ps] = p2; q[2 * ps] = p3;
            }
        }
    }
    __device__ __forceinline__ void tile(int mrow0, int n, const f32x16& acc) {
        float v[16];
        apply(mrow0, n < N ? n : N - 1, acc, v);
        store(mrow0, n, v);
    }
    // the stores of tile(): v = what apply() returned for column min(n, N - 1).  Kernels whose epilogue loads should not queue behind the
    // previous tile's stores call apply() for all their tiles first, then store() for all of them.
    __device__ __forceinline__ void store(int mrow0, int n, float (&v)[16]) {
        const bool col_ok = n < N;
        if (PLANES && !lds_tile && col_ok) store_planes(mrow0, n, v);       // instead of every full-size store to Y below
        if (lds_tile) {          // columns past N hold clamped (finite) duplicates; they are never stored to memory
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow0 + (r & 3) + 8 * (r >> 2);
                if (m < M) lds_tile[m * lds_ld + (n - lds_n0)] = v[r];
            }
        } else if (e.group_max == 16) {
            // max over 16 consecutive columns = the 16 lanes of a DPP row (k_ab = 16 neighbours, layers_pc.py:809-816).  Four
            // v_max_f32 with the DPP modifier per value, four values interleaved per asm block (a DPP read needs two wait states after a
            // write of the same register) -- instead of four ds_bpermute round trips with NaN bookkeeping per value (~40 instructions and four
            // LDS latencies each: the epilogue of a 256-channel layer cost as much as its K loop).  torch.max semantics (NaN propagates):
            // the NaN flags of the lane's 16 values travel as ONE bit mask, OR-ed over the row the same way.
            float mx[16];
            unsigned nanbits = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow0 + (r & 3) + 8 * (r >> 2);
                const bool ok = col_ok && m < M;
                if (!PLANES && e.gmax_out && ok) Y[((long long)b * M + m) * N + n] = v[r];
                mx[r] = ok ? v[r] : -__builtin_inff();
                nanbits |= (mx[r] != mx[r]) ? (1u << r) : 0u;
            }
#define DI2P_OR_DPP(CTRL) nanbits |= (unsigned)__builtin_amdgcn_update_dpp((int)nanbits, (int)nanbits, CTRL, 0xf, 0xf, false)
            DI2P_OR_DPP(0xB1); DI2P_OR_DPP(0x4E); DI2P_OR_DPP(0x141); DI2P_OR_DPP(0x140);
#undef DI2P_OR_DPP
#define DI2P_MAX4(CTRL)                                 \
    "v_max_f32_dpp %0, %0, %0 " CTRL "\n\t"            \
    "v_max_f32_dpp %1, %1, %1 " CTRL "\n\t"            \
    "v_max_f32_dpp %2, %2, %2 " CTRL "\n\t"            \
    "v_max_f32_dpp %3, %3, %3 " CTRL "\n\t"
#pragma unroll
            for (int q = 0; q < 4; ++q)
                asm volatile("s_nop 1\n\t"
                             DI2P_MAX4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                             DI2P_MAX4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                             DI2P_MAX4("row_half_mirror row_mask:0xf bank_mask:0xf")
                             DI2P_MAX4("row_mirror row_mask:0xf bank_mask:0xf")
                             : "+v"(mx[4 * q]), "+v"(mx[4 * q + 1]), "+v"(mx[4 * q + 2]), "+v"(mx[4 * q + 3]));
#undef DI2P_MAX4
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow0 + (r & 3) + 8 * (r >> 2);
                const float val = ((nanbits >> r) & 1u) ? __builtin_nanf("") : mx[r];
                if (col_ok && m < M && (n & 15) == 0) e.gmax_dst[((long long)b * M + m) * (N / 16) + n / 16] = val;
            }
        } else if (e.group_max > 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow0 + (r & 3) + 8 * (r >> 2);
                const bool ok = col_ok && m < M;
                if (!PLANES && e.gmax_out && ok) Y[((long long)b * M + m) * N + n] = v[r];
                float mx = ok ? v[r] : -__builtin_inff();
                for (int o = 1; o < e.group_max; o <<= 1) {      // torch.max semantics: NaN propagates
                    const float ot = __shfl_xor(mx, o);
                    mx = (mx != mx || ot != ot) ? __builtin_nanf("") : fmaxf(mx, ot);
                }
                if (ok && (n % e.group_max) == 0) e.gmax_dst[((long long)b * M + m) * (N / e.group_max) + n / e.group_max] = mx;
            }
        } else if (e.transpose_out) {        // Y[b][n][m]: one float4 per row group (M % 4 == 0)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = mrow0 + 8 * g;
                if (col_ok && m < M) *reinterpret_cast<float4*>(Y + ((long long)b * N + n) * M + m) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
            }
        } else if (!PLANES) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow0 + (r & 3) + 8 * (r >> 2);
                if (col_ok && m < M) Y[((long long)b * M + m) * N + n] = v[r];
            }
        }
    }
};
using EpiPointwise = EpiPointwiseT<-1, -1>;
using EpiPointwisePlanes = EpiPointwiseT<-1, -1, true>;

template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void pointwise_gemm_kernel(SrcDev srcs, const float* __restrict__ Wt, float* __restrict__ Y,
                                                                       int M, int K, int N, EpiDev epi) {
    extern __shared__ float lds[];
    LoaderWt la{Wt, K, M};
    LoaderConcat lb;
    lb.s = srcs;
    lb.b = blockIdx.z;
    lb.N = N;
    lb.K = K;
    EpiPointwise ep{epi, Y, (int)blockIdx.z, M, N};
    mfma_gemm_block<Cfg>(lds, la, lb, ep, K, blockIdx.y * Cfg::BM, blockIdx.x * Cfg::BN);
}

template <class Cfg, bool DENSE, bool DEPTH2 = false>
__global__ __launch_bounds__(Cfg::THREADS) void pointwise_gemm_vec_kernel(SrcDev srcs, const float* __restrict__ Wt, float* __restrict__ Y,
                                                                           int M, int K, int N, EpiDev epi) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    LoaderWt4 la{Wt, K, M};
    LoaderConcat4<DENSE> lb;
    lb.s = srcs;
    lb.b = blockIdx.z;
    lb.N = N;
    lb.K = K;
    EpiPointwise ep{epi, Y, (int)blockIdx.z, M, N};
    if (DEPTH2) mfma_gemm_block_vec2<Cfg>(lds, la, lb, ep, K, blockIdx.y * Cfg::BM, blockIdx.x * Cfg::BN);
    else mfma_gemm_block_vec<Cfg>(lds, la, lb, ep, K, blockIdx.y * Cfg::BM, blockIdx.x * Cfg::BN);
}

// ---- fused per-point head (per_point_pn of networks_united.py:57-74,194-197, coarse variant 736 -> 128 -> 128 -> P):
// layer 0 (dense channels + gathered per-node products), layer 1 and the P-channel output layer in ONE kernel.  A
// workgroup owns 64 points; the 128 x 64 activation tile of layer 0 is written to LDS in operand layout, layer 1 reads
// it from there (mfma_gemm_block_blds) and overwrites it with its own output, and the output layer is a 128-term fma
// chain per (point, class) in the k order of the MFMA -- so the result is BIT-IDENTICAL to the three separate launches,
// while the two 128-channel activations (2 x 335 MB written and read back per 32-frame step) never leave the CU.
struct HeadTail {
    const float* W1t;      // [M][M]   layer-1 weight, k-major
    const float* sc1;
    const float* sh1;
    const float* W2t;      // [M][P]
    const float* sc2;
    const float* sh2;
    int relu1, relu2, P;
};
// K-step 16: 24 KB of staging + the 32 KB activation tile = 56 KB per workgroup (K-step 32: 80 KB).  Alone the two are level; in the
// 8-stream pipeline the smaller footprint is worth +1.3 % frames/s (more workgroups of the other families fit beside it).
#ifndef DI2P_HEAD_BK
#define DI2P_HEAD_BK 16
#endif
using HeadCfg = TileCfg<2, 2, 2, 1, DI2P_HEAD_BK>;       // 128 rows x 64 points, 4 waves of 64 x 32
constexpr int HEAD_M = 128, HEAD_BN = 64;

template <bool G33>       // G33: layer 0 gathers 3 + 3 neighbours (both tables present): compile-time epilogue
__global__ __launch_bounds__(HeadCfg::THREADS) void point_head_kernel(SrcDev srcs, const float* __restrict__ W0t, int K0, EpiDev e0,
                                                                       HeadTail tl, float* __restrict__ out, int N) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* stage = lds;                              // operand staging of the tile engine
    float* hbuf = lds + HeadCfg::LDS_FLOATS;         // [HEAD_M][HEAD_BN] activation tile
    const int b = blockIdx.y, j_blk = blockIdx.x * HEAD_BN;
    {   // layer 0
        LoaderWt4 la{W0t, K0, HEAD_M};
        LoaderConcat4<true> lb;
        lb.s = srcs; lb.b = b; lb.N = N; lb.K = K0;
        EpiPointwiseT<G33 ? 3 : -1, G33 ? 3 : -1> ep{e0, nullptr, b, HEAD_M, N};
        ep.lds_tile = hbuf; ep.lds_n0 = j_blk; ep.lds_ld = HEAD_BN;
        mfma_gemm_block_vec<HeadCfg>(stage, la, lb, ep, K0, 0, j_blk);
    }
    __syncthreads();
    {   // layer 1: B operand = hbuf, output back into hbuf
        LoaderWt4 la{tl.W1t, HEAD_M, HEAD_M};
        EpiDev e1{};
        e1.scale = tl.sc1; e1.shift = tl.sh1; e1.relu = tl.relu1; e1.group_max = 1;
        EpiPointwise ep{e1, nullptr, b, HEAD_M, N};
        ep.lds_tile = hbuf; ep.lds_n0 = j_blk; ep.lds_ld = HEAD_BN;
        mfma_gemm_block_blds<HeadCfg>(stage, la, hbuf, HEAD_BN, ep, HEAD_M, 0, j_blk);
    }
    __syncthreads();
    // output layer: wave p computes class p for the 64 points (k ascending: the order of the MFMA accumulation)
    const int lane = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int n = j_blk + lane;
    if (p < tl.P) {
        float a = 0.0f;
        for (int k0 = 0; k0 < HEAD_M; k0 += 16) {      // operands of 16 steps fetched together, then the ordered fma chain
            float w[16], h[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { w[u] = tl.W2t[(k0 + u) * tl.P + p]; h[u] = hbuf[(k0 + u) * HEAD_BN + lane]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) a = fmaf(w[u], h[u], a);
        }
        if (tl.sc2) a *= tl.sc2[p];
        if (tl.sh2) a += tl.sh2[p];
        if (tl.relu2) a = fmaxf(a, 0.0f);
        if (n < N) out[((long long)b * tl.P + p) * N + n] = a;
    }
}

// ---- fused narrow PointNet chains (first_pointnet 7 -> 32 -> 32 -> 32 and second_pointnet (32 + gathered 32) -> 64 -> 64 of
// networks_pc.py:36-43,60-75): two or three pointwise layers of ONE width M in one launch, WAVE-AUTONOMOUS.  A wave owns all M channels
// of 32 * TN points through the whole chain: the accumulator tile a layer leaves in a lane's registers (16 rows of one column per 32 x 32
// block) IS the next layer's B operand for that column, up to which half-wave holds which row -- lanes l and l + 32 share a column and
// hold rows 8g + {0..3} and 8g + {4..7}; one v_permlane32_swap per register pair puts rows (2s, 2s + 1) into the two halves of one
// register, the operand of K-step s.  No LDS traffic for activations, no workgroup barrier after the weights (9-40 KB, staged to LDS
// once per workgroup) are in place; layer 0 reads its operand rows straight from memory (coalesced 128 B per row and half-wave).
// Same K order, same MFMA sequence and the same epilogue code as the separate launches: BIT-IDENTICAL to them (up to the sign of
// zero: all-zero K-steps of a padded K are skipped), while the hidden activations (2 x 84 MB + 168 MB written and read back per
// 32-frame step) never exist in memory.
struct ChainTail {
    const float* W1t;      // [M][M] k-major
    const float* sc1;
    const float* sh1;
    const float* W2t;      // [M][M] (three-layer chain)
    const float* sc2;
    const float* sh2;
    int relu1, relu2;
};

// v[16] (rows 8g + j + 4*half of one column, r = 4g + j) -> f[16] with f[s] = {row 2s in lanes 0-31, row 2s + 1 in lanes 32-63}
__device__ __forceinline__ void chain_rows_to_ksteps(const float (&v)[16], float* f) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        // swap(X, Y): X' = {X.lo, Y.lo}, Y' = {X.hi, Y.hi}.  X = rows (8g, 8g + 4), Y = rows (8g + 1, 8g + 5) -> X' = rows (8g, 8g + 1), Y' = (8g + 4, 8g + 5)
        auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[4 * g + 0]), __float_as_uint(v[4 * g + 1]), false, false);
        auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[4 * g + 2]), __float_as_uint(v[4 * g + 3]), false, false);
        f[4 * g + 0] = __uint_as_float(p[0]);      // K-step 4g    : rows 8g,     8g + 1
        f[4 * g + 1] = __uint_as_float(q[0]);      // K-step 4g + 1: rows 8g + 2, 8g + 3
        f[4 * g + 2] = __uint_as_float(p[1]);      // K-step 4g + 2: rows 8g + 4, 8g + 5
        f[4 * g + 3] = __uint_as_float(q[1]);      // K-step 4g + 3: rows 8g + 6, 8g + 7
    }
}

// M: width (32 / 64); KS0: K-steps (pairs of input channels) of layer 0, >= ceil(K0 / 2); NL: layers; TN: 32-point blocks per wave and trip
#ifndef DI2P_CHAIN_WAVES
#define DI2P_CHAIN_WAVES 2
#endif
template <int M, int KS0, int NL, int TN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DI2P_CHAIN_WAVES, 8))) void point_chain_kernel(const float* __restrict__ X, long long x_bs, int x_rs, const float* __restrict__ W0t,
                                                           int K0, EpiDev e0, ChainTail tl, float* __restrict__ Y, int N, int nblk, int total) {
    constexpr int TM = M / 32, KS = M / 2, KSMAX = KS0 > KS ? KS0 : KS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w0 = lds;                       // [2 * KS0][M], rows >= K0 zero
    float* w1 = w0 + 2 * KS0 * M;          // [M][M]
    float* w2 = w1 + M * M;
    float* ss = w2 + (NL == 3 ? M * M : 0);      // scale / shift rows of the layers: [6][M] (the epilogues read them from LDS, not through the texture path)
    for (int i = threadIdx.x; i < 2 * KS0 * M; i += 256) w0[i] = i < K0 * M ? W0t[i] : 0.0f;
    for (int i = threadIdx.x; i < M * M; i += 256) w1[i] = tl.W1t[i];
    if (NL == 3)
        for (int i = threadIdx.x; i < M * M; i += 256) w2[i] = tl.W2t[i];
    {
        const float* rows[6] = {e0.scale, e0.shift, tl.sc1, tl.sh1, NL == 3 ? tl.sc2 : nullptr, NL == 3 ? tl.sh2 : nullptr};
#pragma unroll
        for (int q = 0; q < 6; ++q)
            if (rows[q] && threadIdx.x < M) ss[q * M + threadIdx.x] = rows[q][threadIdx.x];
    }
    __syncthreads();
    if (e0.scale) e0.scale = ss;
    if (e0.shift) e0.shift = ss + M;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    EpiDev e1{}, e2{};
    e1.scale = tl.sc1 ? ss + 2 * M : nullptr; e1.shift = tl.sh1 ? ss + 3 * M : nullptr; e1.relu = tl.relu1; e1.group_max = 1;
    e2.scale = tl.sc2 ? ss + 4 * M : nullptr; e2.shift = tl.sh2 ? ss + 5 * M : nullptr; e2.relu = tl.relu2; e2.group_max = 1;

    // The blocks of ALL frames form one list (block g = frame g / nblk, columns (g % nblk) * 32 * TN ...) that the waves of the launch walk
    // with a common stride: the host sizes the grid to whole workgroups per compute unit and equal trip counts, so the weights are staged
    // once per resident workgroup and no wave idles while another finishes.
    // Layer 0's operand rows come straight from memory (rows >= K0: zero; their weights are zero too); the rows of the wave's NEXT block
    // are requested before this block's arithmetic starts.
    float fn[TN][KS0];
    auto request = [&](int g) {
        const int fb = g / nblk, blk = g - fb * nblk;
        const float* Xb = X + (long long)fb * x_bs;
#pragma unroll
        for (int s = 0; s < KS0; ++s) {
            const int k = 2 * s + half;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float x = Xb[(long long)min(k, K0 - 1) * x_rs + min((blk * TN + j) * 32 + l31, N - 1)];
                fn[j][s] = k < K0 ? x : 0.0f;
            }
        }
    };
    const int g0 = blockIdx.x * 4 + wave, stride = gridDim.x * 4;
    if (g0 < total) request(g0);
    for (int g = g0; g < total; g += stride) {
        const int b = g / nblk, blk = g - b * nblk;
        EpiPointwise ep0{e0, nullptr, b, M, N}, ep1{e1, nullptr, b, M, N}, ep2{e2, nullptr, b, M, N};
        int n[TN], nc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) { n[j] = (blk * TN + j) * 32 + l31; nc[j] = min(n[j], N - 1); }
        float f[TN][KSMAX];                 // B operands of the running layer, one register per K-step
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int s = 0; s < KS0; ++s) f[j][s] = fn[j][s];
        request(min(g + stride, total - 1));
        f32x16 acc[TM][TN];
        auto zero = [&]() {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        };
        auto layer = [&](const float* w, auto ks_tag) {
            constexpr int STEPS = decltype(ks_tag)::value;
            const float* wl = w + half * M + l31;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                float a[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = wl[2 * s * M + i * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], f[j][s], acc[i][j], 0, 0, 0);
            }
        };
        auto to_operands = [&](const EpiPointwise& ep) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float v[16];
                    ep.apply(i * 32 + 4 * half, nc[j], acc[i][j], v);
                    chain_rows_to_ksteps(v, &f[j][i * 16]);
                }
        };
        auto to_memory = [&](const EpiPointwise& ep) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float v[16];
                    ep.apply(i * 32 + 4 * half, nc[j], acc[i][j], v);
                    if (n[j] < N) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            Y[((long long)b * M + i * 32 + 4 * half + (r & 3) + 8 * (r >> 2)) * N + n[j]] = v[r];
                    }
                }
        };
        zero();
        layer(w0, std::integral_constant<int, KS0>{});
        to_operands(ep0);
        zero();
        layer(w1, std::integral_constant<int, KS>{});
        if (NL == 3) {
            to_operands(ep1);
            zero();
            layer(w2, std::integral_constant<int, KS>{});
            to_memory(ep2);
        } else {
            to_memory(ep1);
        }
    }
}

// ---- the fused per-point head, wave-autonomous (the design of point_chain_kernel at width 128): a wave owns all 128 channels of 32 points
// through layer 0 (dense channels from memory + gathered node products), layer 1 (its operand = layer 0's accumulators after one
// v_permlane32_swap per register pair) and the P-channel output layer (an ordered 128-term fma chain that alternates between the two
// half-waves holding a column's rows).  ALL weights (K0 <= 96: 48 + 64 + 2 KB) and scale / shift rows sit in LDS for the lifetime of a
// workgroup (one 8-wave workgroup per compute unit, two waves per SIMD), the block list of all frames is walked with a common stride:
// no operand staging, no activation tile in LDS, no barrier after the first.  Same K order, MFMA sequence and epilogue code as
// point_head_kernel and the three separate launches: bit-identical to both.
template <bool G33, int KS0>
__global__ __launch_bounds__(512) void point_head_reg_kernel(SrcDev srcs, const float* __restrict__ W0t, int K0, EpiDev e0, HeadTail tl,
                                                              float* __restrict__ out, int N, int nblk, int total) {
    constexpr int M = HEAD_M, TM = M / 32, KS = M / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w0 = lds;                       // [2 * KS0][M], rows >= K0 zero
    float* w1 = w0 + 2 * KS0 * M;          // [M][M]
    float* w2 = w1 + M * M;                // [M][4]  (P <= 4 output channels, row stride 4)
    float* ss = w2 + M * 4;                // scale0, shift0, scale1, shift1: [4][M]
    for (int i = threadIdx.x; i < 2 * KS0 * M; i += 512) w0[i] = i < K0 * M ? W0t[i] : 0.0f;
    for (int i = threadIdx.x; i < M * M; i += 512) w1[i] = tl.W1t[i];
    for (int i = threadIdx.x; i < M * 4; i += 512) w2[i] = (i & 3) < tl.P ? tl.W2t[(i >> 2) * tl.P + (i & 3)] : 0.0f;
    {
        const float* rows[4] = {e0.scale, e0.shift, tl.sc1, tl.sh1};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (rows[q] && threadIdx.x < M) ss[q * M + threadIdx.x] = rows[q][threadIdx.x];
    }
    __syncthreads();
    if (e0.scale) e0.scale = ss;
    if (e0.shift) e0.shift = ss + M;
    EpiDev e1{};
    e1.scale = tl.sc1 ? ss + 2 * M : nullptr; e1.shift = tl.sh1 ? ss + 3 * M : nullptr; e1.relu = tl.relu1; e1.group_max = 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int c0 = srcs.c_end[0];

    // layer 0's operand rows come straight from memory (rows >= K0: zero; their weights are zero too).  The rows of a wave's NEXT block are
    // requested as soon as this block's layer 0 is done with the registers -- a whole layer 1 (256 matrix instructions) ahead of their use:
    // behind another wave's burst of gather loads in the texture-path queue a request can wait many microseconds.
    constexpr int CH = 8;
    static_assert(KS0 % CH == 0, "layer 0 = whole chunks");
    float f0[KS0];
    // (buffer loads: the frame's base in a descriptor, the K-step's row offset in a scalar register, one per-lane byte offset per source --
    //  with 64-bit per-lane addresses the 48 requests of a block cost 96 address registers and 400 B of scratch per lane.  c0 is even
    //  (host-checked): both half-waves of a K-step read the same source; rows past a source's end read as zero.)
    auto request = [&](int g) {
        const int fb = __builtin_amdgcn_readfirstlane(g / nblk);
        const int col = min((g - fb * nblk) * 32 + l31, N - 1);
        const int rs0 = srcs.row_stride[0], rs1 = srcs.row_stride[1];
        const auto r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(srcs.ptr[0] + (long long)fb * srcs.batch_stride[0]), 0, c0 * rs0 * 4, 0x00020000);
        const auto r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(srcs.ptr[1] + (long long)fb * srcs.batch_stride[1]), 0, (K0 - c0) * rs1 * 4, 0x00020000);
        const int v0 = (col + half * rs0) * 4, v1 = (col + half * rs1) * 4;
#pragma unroll
        for (int s = 0; s < KS0; ++s) {
            if (2 * s < c0) f0[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r0, v0, 2 * s * rs0 * 4, 0));
            else f0[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, v1, (2 * s - c0) * rs1 * 4, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const int g_first = blockIdx.x * 8 + wave, stride = gridDim.x * 8;
    if (g_first < total) request(g_first);
    for (int g = g_first; g < total; g += stride) {
        const int b = g / nblk, blk = g - b * nblk;
        const int n = blk * 32 + l31, nc = min(n, N - 1);
        using Epi0 = EpiPointwiseT<G33 ? 3 : -1, G33 ? 3 : -1>;
        Epi0 ep0{e0, nullptr, b, M, N};
        typename Epi0::Gathered G;
        if (G33) ep0.gather_setup(nc, G);
        f32x16 acc[TM];
        auto zero = [&]() {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        };
        // K-steps k0 .. k0 + STEPS - 1 of a layer on the operands fs[0 .. STEPS): one scheduling region with an explicit software pipeline, the
        // weight-fragment reads of K-step s + 3 in front of the matrix instructions of K-step s (12 LDS reads in flight: the wait counter
        // holds 15).  Left alone, hipcc hoists every read of a layer to its top (1.2 KB of scratch per lane).
        auto steps = [&](const float* w, int k0, const float* fs, auto n_tag) {
            constexpr int STEPS = decltype(n_tag)::value;
            const float* wl = w + half * M + l31;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                float a[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = wl[2 * (k0 + s) * M + i * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], fs[s], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 3 * TM, 0);
#pragma unroll
            for (int s = 0; s < STEPS - 3; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * TM, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        zero();
#pragma unroll
        for (int s0 = 0; s0 < KS0; s0 += CH) steps(w0, s0, &f0[s0], std::integral_constant<int, CH>{});
        float f1[KS];                       // layer 1's operands: layer 0's output, one register per K-step
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float v[16], t16[16];
            if (G33) ep0.gather_sum(G, i * 32 + 4 * half, t16);
            ep0.apply_pre(i * 32 + 4 * half, nc, acc[i], G33 ? t16 : nullptr, v);
            chain_rows_to_ksteps(v, &f1[i * 16]);
        }
        __builtin_amdgcn_sched_barrier(0);
        request(min(g + stride, total - 1));
        zero();
        steps(w1, 0, f1, std::integral_constant<int, KS>{});
        float h[TM][16];                    // layer 1's output: rows i * 32 + 8q + j + 4 * half of this lane's column (r = 4q + j)
        {
            EpiPointwise ep1{e1, nullptr, b, M, N};
#pragma unroll
            for (int i = 0; i < TM; ++i) ep1.apply(i * 32 + 4 * half, nc, acc[i], h[i]);
        }
        // output layer: k ascending, the order of the MFMA accumulation.  Rows 8q' .. 8q' + 3 of a column live in lane l, rows 8q' + 4 .. + 7
        // in lane l + 32: the running sum visits the low half-wave, is handed over (permlane32_swap of a register with itself broadcasts one
        // half-wave's value to both), visits the high half-wave, and is handed back.
        for (int p = 0; p < tl.P; ++p) {
            const float* wp = w2 + 16 * half + p;
            float a = 0.0f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = wp[(i * 32 + 8 * q + j) * 4];
                    float t = a;            // low half-wave: rows 8q' + 0..3
#pragma unroll
                    for (int j = 0; j < 4; ++j) t = fmaf(w[j], h[i][4 * q + j], t);
                    const float lo = __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false)[0]);
                    float u = lo;           // high half-wave: rows 8q' + 4..7
#pragma unroll
                    for (int j = 0; j < 4; ++j) u = fmaf(w[j], h[i][4 * q + j], u);
                    a = __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(u), __float_as_uint(u), false, false)[1]);
                }
            if (tl.sc2) a *= tl.sc2[p];
            if (tl.sh2) a += tl.sh2[p];
            if (tl.relu2) a = fmaxf(a, 0.0f);
            if (n < N && half == 0) out[((long long)b * tl.P + p) * N + n] = a;
        }
    }
}

// ---- attention pooling: out[b,c,m] = (1/HW) sum_hw feat[b,c,hw] * score[b,hw,m]
struct LoaderFeat {  // A[k=hw][m=c] = feat[b][c][hw]
    const float* feat;
    int HW, C;
    __device__ __forceinline__ float load(int k, int m) const { return (k < HW && m < C) ? feat[(long long)m * HW + k] : 0.0f; }
};
struct LoaderScore {
    const float* score;  // [HW][Mn] of this batch
    int HW, Mn, n;
    bool valid;
    __device__ __forceinline__ void column(int j) { n = j; valid = j < Mn; }
    __device__ __forceinline__ void begin_tile(int) {}
    __device__ __forceinline__ float load(int k) const { return (valid && k < HW) ? score[(long long)k * Mn + n] : 0.0f; }
};
struct EpiMean {
    float* out;  // [C][Mn] of this batch
    int C, Mn;
    float inv_div;
    __device__ __forceinline__ void tile(int mrow0, int n, const f32x16& acc) {
        if (n >= Mn) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + (r & 3) + 8 * (r >> 2);
            if (m < C) out[(long long)m * Mn + n] = acc[r] / inv_div;
        }
    }
};
template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void attention_pool_kernel(const float* __restrict__ feat, const float* __restrict__ score,
                                                                       float* __restrict__ out, int C, int HW, int Mn) {
    extern __shared__ float lds[];
    const int b = blockIdx.z;
    LoaderFeat la{feat + (long long)b * C * HW, HW, C};
    LoaderScore lb{score + (long long)b * HW * Mn, HW, Mn, 0, false};
    EpiMean ep{out + (long long)b * C * Mn, C, Mn, (float)HW};
    mfma_gemm_block<Cfg>(lds, la, lb, ep, HW, blockIdx.y * Cfg::BM, blockIdx.x * Cfg::BN);
}

// out[b][m] = sum_k Wt[k0+k][m] * v[b][k]  (+ the same for a second (k1, v1) pair).  One block = 64 output channels x 4
// k-slices (one wave per slice: its v[b][k] reads are wave-uniform, its Wt reads 256 contiguous bytes); 4 independent
// accumulators per lane keep the dependent FMA chain at Kv/16, the slices are combined through LDS.  With two pairs the
// result is (gemv0) + (gemv1), each summed exactly as a single-pair call would: node_b_pn's two broadcast inputs
// (networks_united.py:152-155) become ONE launch instead of two launches and an elementwise add.
__device__ __forceinline__ float gemv_slice(const float* __restrict__ Wt, int M, int mc, int k0, const float* __restrict__ vb, int Kv, int slice) {
    const int per = (Kv + 3) / 4;
    const int kb = slice * per, ke = min(kb + per, Kv);
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int k = kb;
    for (; k + 3 < ke; k += 4) {
        a0 += Wt[(long long)(k0 + k) * M + mc] * vb[k];
        a1 += Wt[(long long)(k0 + k + 1) * M + mc] * vb[k + 1];
        a2 += Wt[(long long)(k0 + k + 2) * M + mc] * vb[k + 2];
        a3 += Wt[(long long)(k0 + k + 3) * M + mc] * vb[k + 3];
    }
    for (; k < ke; ++k) a0 += Wt[(long long)(k0 + k) * M + mc] * vb[k];
    return (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(256) void batch_gemv_kernel(const float* __restrict__ Wt, int M, int k0, const float* __restrict__ v,
                                                         int Kv, int k1, const float* __restrict__ v1, int Kv1,
                                                         float* __restrict__ out) {
    __shared__ float part[2][4][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int m = blockIdx.x * 64 + lane;
    const int mc = min(m, M - 1);
    part[0][slice][lane] = gemv_slice(Wt, M, mc, k0, v + (long long)b * Kv, Kv, slice);
    if (v1) part[1][slice][lane] = gemv_slice(Wt, M, mc, k1, v1 + (long long)b * Kv1, Kv1, slice);
    __syncthreads();
    if (slice == 0 && m < M) {
        float r = (part[0][0][lane] + part[0][1][lane]) + (part[0][2][lane] + part[0][3][lane]);
        if (v1) r += (part[1][0][lane] + part[1][1][lane]) + (part[1][2][lane] + part[1][3][lane]);
        out[(long long)b * M + m] = r;
    }
}

template <class Cfg>
void launch_pw(const SrcDev& s, const float* Wt, float* Y, int B, int M, int K, int N, const EpiDev& e, hipStream_t st) {
    const dim3 grid(di2p_cdiv(N, Cfg::BN), di2p_cdiv(M, Cfg::BM), B);
    hipLaunchKernelGGL(pointwise_gemm_kernel<Cfg>, grid, dim3(Cfg::THREADS), Cfg::LDS_FLOATS * sizeof(float), st, s, Wt, Y, M, K, N, e);
}

template <class Cfg>
void launch_pw_vec(bool dense, const SrcDev& s, const float* Wt, float* Y, int B, int M, int K, int N, const EpiDev& e, hipStream_t st) {
    const dim3 grid(di2p_cdiv(N, Cfg::BN), di2p_cdiv(M, Cfg::BM), B);
    const bool d2 = di2p_opt(DI2P_OPT_CONV_DEPTH1) == 0 && K > 2 * Cfg::BK;      // depth-2 prefetch pays from three K-steps on
    if (dense && d2)
        hipLaunchKernelGGL((pointwise_gemm_vec_kernel<Cfg, true, true>), grid, dim3(Cfg::THREADS), Cfg::LDS_FLOATS * sizeof(float), st, s, Wt, Y, M, K, N, e);
    else if (dense)
        hipLaunchKernelGGL((pointwise_gemm_vec_kernel<Cfg, true>), grid, dim3(Cfg::THREADS), Cfg::LDS_FLOATS * sizeof(float), st, s, Wt, Y, M, K, N, e);
    else if (d2)
        hipLaunchKernelGGL((pointwise_gemm_vec_kernel<Cfg, false, true>), grid, dim3(Cfg::THREADS), Cfg::LDS_FLOATS * sizeof(float), st, s, Wt, Y, M, K, N, e);
    else
        hipLaunchKernelGGL((pointwise_gemm_vec_kernel<Cfg, false>), grid, dim3(Cfg::THREADS), Cfg::LDS_FLOATS * sizeof(float), st, s, Wt, Y, M, K, N, e);
}


// ----------------------------------------------------------------------------------------------------------------------------------
// "bf16x3": fp32 contractions through bf16 matrix instructions with an EXACT three-way operand split.
//   x = x1 + x2 + x3 (three truncated bf16 terms of 8 significand bits each: 24 bits, nothing is lost), and
//   a * b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) + terms below 2^-24 |a| |b|        -- six bf16 products, fp32 accumulation,
// smallest terms first.  v_mfma_f32_32x32x16_bf16 does K = 16 in 32 cycles where v_mfma_f32_32x32x2_f32 needs 8 x 64: six of them are
// 2.67x the fp32-MFMA rate.  Against an fp64 contraction the result is as accurate as the fp32-MFMA kernel's (the dropped terms are below the
// rounding of the fp32 accumulation; tests/test_gpu_contractions.py asserts err_bf16x3 <= err_fp32mfma on the golden operands).
// Used for the GEMM-shaped layers (K >= 128, M % 128 == 0: the kNN-fusion layers and the node-level PointNets); narrow point layers are
// memory-bound and stay on the fp32 kernels.  Non-finite inputs: x = +-inf splits into (inf, NaN, NaN), i.e. the output is NaN where the
// fp32 kernel gives +-inf or NaN -- non-finite either way.
//   weights: split ONCE (di2p_bf16x3_pack) into [Kp/8][Mp][3] x 8 bf16 (Kp = K rounded up to 32, Mp = M rounded up to 128, zero filled) and read
//            straight from L2 as MFMA A fragments;
//   activations: fp32 in memory, loaded through the SAME loaders as the fp32 kernels and split while they are staged into LDS
//            (5.5 vector instructions per value); LDS layout [plane][k-group of 8][k-half][column] x 8 bytes: a thread stores 32 contiguous
//            bytes per plane (its 4 columns x 4 consecutive k), a fragment read is two conflict-free 8-byte reads.
// Workgroup 128 x 128, 4 waves of 64 x 64 (2 x 2 MFMA tiles), K-step 32, 48 KB of LDS, two workgroups per CU.
constexpr int X3_BM = 128, X3_BN = 128, X3_BK = 32, X3_KG = X3_BK / 8;

template <bool DENSE, bool PLANES = false>
__global__ __launch_bounds__(256, 2) void pointwise_gemm_x3_kernel(SrcDev srcs, const u32x4_t* __restrict__ Wp, float* __restrict__ Y, int M, int K,
                                                                    int N, int Mp, EpiDev epi) {
    __shared__ __attribute__((aligned(16))) u32x2_t Bs[2][3][X3_KG][2][X3_BN];        // 2 x 24 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m_blk = blockIdx.y * X3_BM, n_blk = blockIdx.x * X3_BN, b = blockIdx.z;
    const int T = (K + X3_BK - 1) / X3_BK;
    LoaderConcat4<DENSE> lb;
    lb.s = srcs; lb.b = b; lb.N = N; lb.K = K;
    // staging role: 4 columns (tid & 31), 4 consecutive k: k-group (tid >> 6), half (tid >> 5) & 1
    const int cq = tid & 31, skg = tid >> 6, shh = (tid >> 5) & 1;
    lb.column4(n_blk + 4 * cq);
    float4 st[4];
    auto gload = [&](int t) {
        const int k0 = t * X3_BK;
        lb.begin_tile(k0);
#pragma unroll
        for (int r = 0; r < 4; ++r) st[r] = lb.load4(k0 + skg * 8 + shh * 4 + r);      // rows k >= K re-read row K-1 and meet zero weights
    };
    auto sstore = [&](int buf) {
        u32x2_t p1[4], p2[4], p3[4];
        x3_split4(st[0].x, st[1].x, st[2].x, st[3].x, p1[0], p2[0], p3[0]);
        x3_split4(st[0].y, st[1].y, st[2].y, st[3].y, p1[1], p2[1], p3[1]);
        x3_split4(st[0].z, st[1].z, st[2].z, st[3].z, p1[2], p2[2], p3[2]);
        x3_split4(st[0].w, st[1].w, st[2].w, st[3].w, p1[3], p2[3], p3[3]);
        u32x4_t* d1 = reinterpret_cast<u32x4_t*>(&Bs[buf][0][skg][shh][4 * cq]);
        u32x4_t* d2 = reinterpret_cast<u32x4_t*>(&Bs[buf][1][skg][shh][4 * cq]);
        u32x4_t* d3 = reinterpret_cast<u32x4_t*>(&Bs[buf][2][skg][shh][4 * cq]);
        d1[0] = u32x4_t{p1[0].x, p1[0].y, p1[1].x, p1[1].y}; d1[1] = u32x4_t{p1[2].x, p1[2].y, p1[3].x, p1[3].y};
        d2[0] = u32x4_t{p2[0].x, p2[0].y, p2[1].x, p2[1].y}; d2[1] = u32x4_t{p2[2].x, p2[2].y, p2[3].x, p2[3].y};
        d3[0] = u32x4_t{p3[0].x, p3[0].y, p3[1].x, p3[1].y}; d3[1] = u32x4_t{p3[2].x, p3[2].y, p3[3].x, p3[3].y};
    };
    // A fragments straight from memory: [Kp/8][Mp][3] x 16 bytes; lane = row (l31), k-group (half)
    u32x4_t af[2][2][3];                                     // [stage][tile i][plane]
    auto aload = [&](int kg_global, int stg) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32x4_t* p = Wp + ((long long)(kg_global + half) * Mp + m_blk + wm * 64 + i * 32 + l31) * 3;
#pragma unroll
            for (int q = 0; q < 3; ++q) af[stg][i][q] = p[q];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // one k16 sub-step: B fragments from LDS, the A fragments of the NEXT sub-step requested first
    auto substep = [&](int buf, int sub, int next_kg, bool has_next) __attribute__((always_inline)) {
        if (has_next) aload(next_kg, sub ^ 1);
        u32x4_t bf[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int n = wn * 64 + j * 32 + l31;
                const u32x2_t lo = Bs[buf][q][2 * sub + half][0][n], hi = Bs[buf][q][2 * sub + half][1][n];
                bf[j][q] = u32x4_t{lo.x, lo.y, hi.x, hi.y};
            }
        DI2P_MFMA_BEGIN();
        // smallest terms first; four independent accumulators between two matrix instructions of one chain
#define DI2P_X3_PROD(QA, QB)                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                                      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[sub][i][QA]), __builtin_bit_cast(bf16x8_t, bf[j][QB]), acc[i][j], 0, 0, 0);
        DI2P_X3_PROD(2, 0) DI2P_X3_PROD(1, 1) DI2P_X3_PROD(0, 2)
        DI2P_X3_PROD(1, 0) DI2P_X3_PROD(0, 1)
        DI2P_X3_PROD(0, 0)
#undef DI2P_X3_PROD
        DI2P_MFMA_END();
    };
    gload(0);
    sstore(0);
    aload(0, 0);
    __syncthreads();
    // steady state without branches (a branch arm without loads turns every wait of the other arm into vmcnt(0))
    for (int t = 0; t + 1 < T; ++t) {
        const int buf = t & 1;
        gload(t + 1);
        substep(buf, 0, t * X3_KG + 2, true);
        substep(buf, 1, t * X3_KG + 4, true);
        sstore(buf ^ 1);
        __syncthreads();
    }
    substep((T - 1) & 1, 0, (T - 1) * X3_KG + 2, true);
    substep((T - 1) & 1, 1, 0, false);
    // every tile's operand loads (gathered rows, scale, shift: 16 bytes each) and arithmetic first, then every tile's stores: loads that follow
    // a store to memory the compiler cannot tell apart are not moved above it -- tile by tile, each tile's round trips came one after the other
    EpiPointwiseT<-1, -1, PLANES> ep{epi, Y, b, M, N};
    float v[2][2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            ep.template apply_pre<true>(m_blk + wm * 64 + i * 32 + 4 * half, min(n_blk + wn * 64 + j * 32 + l31, N - 1), acc[i][j], nullptr, v[i][j]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) ep.store(m_blk + wm * 64 + i * 32 + 4 * half, n_blk + wn * 64 + j * 32 + l31, v[i][j]);
}

// The same contraction with the activations given ALREADY SPLIT: three bf16 planes [B][3][K/4][N] x 8 bytes (4 consecutive k of one column),
// what a PLANES epilogue of the producing layer wrote -- the LDS layout of the kernel above, row for row.  A K-step is 24 rows of 1 KB
// (plane, k-quad): six per wave, one 16-byte load and one 16-byte LDS store per lane and row, no arithmetic on the way (the split costs the
// fp32-source kernel 88 of its 225 vector instructions per K-step and wave, the generic concatenating loader's addressing about 80 more, and
// every 128-row workgroup of a column tile repeats both: 52 are left here).  Same A fragments, same products in the same order: bit-identical
// to the fp32-source kernel on the same values.  Needs K % 32 == 0 and N % 128 == 0 (host-checked).
// What the K loop is bound by (round 6, profiles/r06_c25...c29: K sweeps of this kernel with parts of the loop removed): NOT the vector
// instructions (this kernel's loop runs at the fp32-source kernel's rate), not the barrier (removed: no change), not the request order or
// distance (weight fragments in front of the plane rows, or a whole K-step ahead in a second register set: no change), not the fragments'
// 48-byte interleave (plane-major weights: no change), not ds_read2_b64 against ds_read_b128 fragments (a 16-byte plane layout: no change);
// it is the vector-memory path itself: without the weight-fragment requests the loop is 24 % faster, without the plane rows 13 %, without both
// 2.4x (1.65 PFLOP/s of bf16 products), with the fragments coming from an L1-resident range 13 %: 144 KB per K-step and compute unit (the
// three planes make every operand byte count three times) against 3072 matrix cycles per SIMD.
template <bool PLANES>
__global__ __launch_bounds__(256, 2) void pointwise_gemm_x3p_kernel(const u32x4_t* __restrict__ P, const u32x4_t* __restrict__ Wp, float* __restrict__ Y,
                                                                     int M, int K, int N, int Mp, EpiDev epi) {
    __shared__ __attribute__((aligned(16))) u32x2_t Bs[2][3][X3_KG][2][X3_BN];        // 2 x 24 KB: [buffer][24 rows of 1 KB]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m_blk = blockIdx.y * X3_BM, n_blk = blockIdx.x * X3_BN, b = blockIdx.z;
    const int T = K / X3_BK;
    // staging role: rows wave * 6 .. + 5 of the 24 (row = plane * 8 + k-quad of the K-step), columns 2 * lane, 2 * lane + 1
    const int kq = K >> 2, nh = N >> 1;                      // k-quads per plane; 16-byte elements per row
    const u32x4_t* Pf = P + (long long)b * 3 * kq * nh + (n_blk >> 1) + lane;
    int roff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int row = wave * 6 + i;
        roff[i] = ((row >> 3) * kq + (row & 7)) * nh;
    }
    u32x4_t st[6];
    auto gload = [&](int t) {
        const u32x4_t* pt = Pf + (long long)t * 8 * nh;
#pragma unroll
        for (int i = 0; i < 6; ++i) st[i] = pt[roff[i]];
    };
    auto sstore = [&](int buf) {
        u32x4_t* d = reinterpret_cast<u32x4_t*>(&Bs[buf][0][0][0][0]) + wave * 6 * 64 + lane;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i * 64] = st[i];
    };
    u32x4_t af[2][2][3];                                     // [stage][tile i][plane]
    auto aload = [&](int kg_global, int stg) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32x4_t* p = Wp + ((long long)(kg_global + half) * Mp + m_blk + wm * 64 + i * 32 + l31) * 3;
#pragma unroll
            for (int q = 0; q < 3; ++q) af[stg][i][q] = p[q];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    auto substep = [&](int buf, int sub, int next_kg, bool has_next) __attribute__((always_inline)) {
        if (has_next) aload(next_kg, sub ^ 1);
        u32x4_t bf[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int n = wn * 64 + j * 32 + l31;
                const u32x2_t lo = Bs[buf][q][2 * sub + half][0][n], hi = Bs[buf][q][2 * sub + half][1][n];
                bf[j][q] = u32x4_t{lo.x, lo.y, hi.x, hi.y};
            }
        DI2P_MFMA_BEGIN();
#define DI2P_X3_PROD(QA, QB)                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                                      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[sub][i][QA]), __builtin_bit_cast(bf16x8_t, bf[j][QB]), acc[i][j], 0, 0, 0);
        DI2P_X3_PROD(2, 0) DI2P_X3_PROD(1, 1) DI2P_X3_PROD(0, 2)
        DI2P_X3_PROD(1, 0) DI2P_X3_PROD(0, 1)
        DI2P_X3_PROD(0, 0)
#undef DI2P_X3_PROD
        DI2P_MFMA_END();
    };
    gload(0);
    sstore(0);
    aload(0, 0);
    __syncthreads();
    for (int t = 0; t + 1 < T; ++t) {
        const int buf = t & 1;
        gload(t + 1);
        substep(buf, 0, t * X3_KG + 2, true);
        substep(buf, 1, t * X3_KG + 4, true);
        sstore(buf ^ 1);
        __syncthreads();
    }
    substep((T - 1) & 1, 0, (T - 1) * X3_KG + 2, true);
    substep((T - 1) & 1, 1, 0, false);
    // every tile's operand loads (gathered rows, scale, shift: 16 bytes each) and arithmetic first, then every tile's stores: loads that follow
    // a store to memory the compiler cannot tell apart are not moved above it -- tile by tile, each tile's round trips came one after the other
    EpiPointwiseT<-1, -1, PLANES> ep{epi, Y, b, M, N};
    float v[2][2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            ep.template apply_pre<true>(m_blk + wm * 64 + i * 32 + 4 * half, min(n_blk + wn * 64 + j * 32 + l31, N - 1), acc[i][j], nullptr, v[i][j]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) ep.store(m_blk + wm * 64 + i * 32 + 4 * half, n_blk + wn * 64 + j * 32 + l31, v[i][j]);
}

// The planes-source kernel with BOTH operands staged through LDS: a 256 x 128 tile, eight waves (4 x 2 of 64 x 64), one workgroup per
// compute unit.  pointwise_gemm_x3p_kernel is bound by the vector-memory path (see its header: 144 KB per K-step and compute unit, the
// weight fragments requested by both waves that share their rows); here a K-step moves the 256 rows' fragments ONCE (48 KB: four k-groups of
// 12 contiguous KB, copied verbatim -- a fragment read is one ds_read_b128 at 48 bytes per lane, conflict-free) and the plane rows once per
// 256 output rows instead of once per 128 (24 KB): 72 KB per K-step and compute unit for the same matrix work.  144 KB of LDS.
// Same products in the same order per output element: bit-identical to the other two kernels.  Needs M % 256 == 0 on top of K % 32, N % 128.
// Measured (profiles/r06_c30...c34): 5-8 % off the K loop (chain of the three kNN-fusion layers 323 -> 315 us), NOT the halving the byte count
// suggests.  The same loop with its nine requests per wave as LDS-DMA (global_load_lds_dwordx4: no staging registers, no LDS stores), or with
// those issued one by one between the product groups: the same time again.  Stamps in the loop: of ~8.3 k cycles per K-step and wave 1.5 k are
// the wave's own matrix instructions; ~1.9 k go to ISSUING nine requests (the wave stalls at issue), 1.3 k waiting for them, 2.1 k at the barrier
// for the slower waves.  Memory side: an L2 read takes 437 cycles on average (TCP_TCC_READ_REQ_LATENCY / REQ), the L1 is stalled on pending
// requests 43 % of the time, the L2 channels are busy 82 %: each compute unit moves its 576 lines per K-step through a bounded number of
// outstanding L1 misses.  What would change the regime is a tile with 2-4x the matrix work per staged byte (the 256 x 256 / one wave per SIMD
// GEMM of cdna_hip_programming.md) -- not built: the three layers are 0.32 ms of a 5.8 ms step.
constexpr int X3P8_A_EL = 4 * 256 * 3, X3P8_B_EL = 24 * 64, X3P8_BUF_EL = X3P8_A_EL + X3P8_B_EL;      // 16-byte elements per buffer (72 KB)
constexpr int X3P8_LDS_BYTES = 2 * X3P8_BUF_EL * 16;

template <bool PLANES>
__global__ __launch_bounds__(512, 1) void pointwise_gemm_x3p8_kernel(const u32x4_t* __restrict__ P, const u32x4_t* __restrict__ Wp, float* __restrict__ Y,
                                                                      int M, int K, int N, int Mp, EpiDev epi) {
    extern __shared__ __attribute__((aligned(16))) u32x4_t lds8[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m_blk = blockIdx.y * 256, n_blk = blockIdx.x * X3_BN, b = blockIdx.z;
    const int T = K / X3_BK;
    const int kq = K >> 2, nh = N >> 1;                      // k-quads per plane; 16-byte elements per plane row
    // plane rows: this thread moves k-quad `wave` of each of the three planes, columns 2 * lane, 2 * lane + 1
    const u32x4_t* Pf = P + (long long)b * 3 * kq * nh + (long long)wave * nh + (n_blk >> 1) + lane;
    // weight fragments: elements i * 512 + tid (i = 0..5) of the K-step's 4 x 768 (k-group, 256 rows x 3 planes)
    const u32x4_t* Wf = Wp + (long long)m_blk * 3;
    int aoff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int e = i * 512 + tid, kg = e / 768;
        aoff[i] = kg * Mp * 3 + (e - kg * 768);
    }
    u32x4_t sa[6], sb[3];
    auto gload = [&](int t) {
        const u32x4_t* wt = Wf + (long long)t * 4 * Mp * 3;
#pragma unroll
        for (int i = 0; i < 6; ++i) sa[i] = wt[aoff[i]];
        const u32x4_t* pt = Pf + (long long)t * 8 * nh;
#pragma unroll
        for (int i = 0; i < 3; ++i) sb[i] = pt[(long long)i * kq * nh];
    };
    auto sstore = [&](int buf) {
        u32x4_t* d = lds8 + buf * X3P8_BUF_EL;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i * 512 + tid] = sa[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) d[X3P8_A_EL + (i * 8 + wave) * 64 + lane] = sb[i];
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    auto substep = [&](int buf, int sub) __attribute__((always_inline)) {
        const u32x4_t* d = lds8 + buf * X3P8_BUF_EL;
        const u32x2_t* bq = reinterpret_cast<const u32x2_t*>(d + X3P8_A_EL);
        u32x4_t af[2][3], bf[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) af[i][q] = d[((2 * sub + half) * 256 + wm * 64 + i * 32 + l31) * 3 + q];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int n = wn * 64 + j * 32 + l31;
                const u32x2_t lo = bq[(q * 8 + (2 * sub + half) * 2) * X3_BN + n], hi = bq[(q * 8 + (2 * sub + half) * 2 + 1) * X3_BN + n];
                bf[j][q] = u32x4_t{lo.x, lo.y, hi.x, hi.y};
            }
        DI2P_MFMA_BEGIN();
#define DI2P_X3_PROD(QA, QB)                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                                      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i][QA]), __builtin_bit_cast(bf16x8_t, bf[j][QB]), acc[i][j], 0, 0, 0);
        DI2P_X3_PROD(2, 0) DI2P_X3_PROD(1, 1) DI2P_X3_PROD(0, 2)
        DI2P_X3_PROD(1, 0) DI2P_X3_PROD(0, 1)
        DI2P_X3_PROD(0, 0)
#undef DI2P_X3_PROD
        DI2P_MFMA_END();
    };
    gload(0);
    sstore(0);
    __syncthreads();
    for (int t = 0; t + 1 < T; ++t) {
        const int buf = t & 1;
        gload(t + 1);
        substep(buf, 0);
        substep(buf, 1);
        sstore(buf ^ 1);
        __syncthreads();
    }
    substep((T - 1) & 1, 0);
    substep((T - 1) & 1, 1);
    EpiPointwiseT<-1, -1, PLANES> ep{epi, Y, b, M, N};
    float v[2][2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            ep.template apply_pre<true>(m_blk + wm * 64 + i * 32 + 4 * half, min(n_blk + wn * 64 + j * 32 + l31, N - 1), acc[i][j], nullptr, v[i][j]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) ep.store(m_blk + wm * 64 + i * 32 + 4 * half, n_blk + wn * 64 + j * 32 + l31, v[i][j]);
}

// Wt f32 [K][M] (k-major, what the fp32 kernels read) -> [Kp/8][Mp][3][8] bf16, zero filled outside K x M.  One thread per (k-group, m).
// MODE 0: Wt is the [K][M] matrix itself.  MODE 1 / 2: Wt is a 3 x 3 filter bank W[Cout][Cin][3][3] and the matrix is its tap-major form for
// di2p_conv3x3_x3 -- 1: the forward filter, row (tap, ci), column co (K = 9 Cin, M = Cout, ci_n = Cin); 2: the filter of the INPUT GRADIENT,
// flipped and channel-transposed, row (tap, co), column ci (K = 9 Cout, M = Cin, ci_n = Cout): what the training step otherwise builds with a
// flip, a transpose and two copies before every pack.
template <int MODE>
__global__ __launch_bounds__(256) void bf16x3_pack_kernel(const float* __restrict__ Wt, unsigned short* __restrict__ Wp, int K, int M, int Kp, int Mp, int ci_n) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)(Kp / 8) * Mp) return;
    const int kg = (int)(t / Mp), m = (int)(t - (long long)kg * Mp);
    unsigned short* d = Wp + t * 24;
    const long long ps = 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = kg * 8 + i;
        float a = 0.0f;
        if (k < K && m < M) {
            if (MODE == 0) a = Wt[(long long)k * M + m];
            else {
                const int tap = k / ci_n, c = k - tap * ci_n;
                a = MODE == 1 ? Wt[((long long)m * ci_n + c) * 9 + tap] : Wt[((long long)c * M + m) * 9 + 8 - tap];
            }
        }
        const float a1 = x3_hi16(a), r1 = a - a1, a2 = x3_hi16(r1), r2 = r1 - a2;
        d[0 * ps + i] = (unsigned short)(__builtin_bit_cast(unsigned, a1) >> 16);
        d[1 * ps + i] = (unsigned short)(__builtin_bit_cast(unsigned, a2) >> 16);
        d[2 * ps + i] = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
    }
}

// The 3 x 3 filter-bank forms with one thread per (eight input channels, output channel m) and ALL nine taps: the thread reads its 72 values as
// whole runs of memory (forward: 288 contiguous bytes; input gradient: eight runs of 36 bytes that are contiguous ACROSS the lanes' m) and writes nine
// 48-byte entries, contiguous across lanes.  (Per (k-group, m) as in the kernel above every lane touched 3 cache lines for 32 useful bytes: 22 us
// per 512 x 512 bank, 0.72 ms of a training step.)  ci_n % 8 == 0.
template <bool DGRAD>
__global__ __launch_bounds__(256) void bf16x3_pack_conv_kernel(const float* __restrict__ W, unsigned short* __restrict__ Wp, int M, int Mp, int ci_n) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c8n = ci_n >> 3;
    if (t >= (long long)c8n * Mp) return;
    const int c8 = (int)(t / Mp), m = (int)(t - (long long)c8 * Mp);
    float v[8][9];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int c = c8 * 8 + i;
            v[i][tap] = m < M ? (DGRAD ? W[((long long)c * M + m) * 9 + 8 - tap] : W[((long long)m * ci_n + c) * 9 + tap]) : 0.0f;
        }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float r[8], q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { r[i] = v[i][tap] - x3_hi16(v[i][tap]); q[i] = r[i] - x3_hi16(r[i]); }
        u32x4_t* d = reinterpret_cast<u32x4_t*>(Wp + ((long long)(tap * c8n + c8) * Mp + m) * 24);          // three 16-byte stores per entry
        d[0] = u32x4_t{x3_pack_hi(v[0][tap], v[1][tap]), x3_pack_hi(v[2][tap], v[3][tap]), x3_pack_hi(v[4][tap], v[5][tap]), x3_pack_hi(v[6][tap], v[7][tap])};
        d[1] = u32x4_t{x3_pack_hi(r[0], r[1]), x3_pack_hi(r[2], r[3]), x3_pack_hi(r[4], r[5]), x3_pack_hi(r[6], r[7])};
        d[2] = u32x4_t{x3_pack_hi(q[0], q[1]), x3_pack_hi(q[2], q[3]), x3_pack_hi(q[4], q[5]), x3_pack_hi(q[6], q[7])};
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Sources i >= n_src become aliases of source 0 with an empty channel range [K, K): addressable, never selected.
inline void alias_absent_sources(SrcDev& s, int n_src) {
    for (int i = n_src; i < DI2P_MAX_SRC; ++i) {
        s.ptr[i] = s.ptr[0]; s.gidx[i] = s.gidx[0]; s.batch_stride[i] = s.batch_stride[0]; s.row_stride[i] = s.row_stride[0];
        s.mode[i] = s.mode[0]; s.group[i] = s.group[0];
    }
}

}  // namespace

#ifndef DI2P_CHAIN_TN32
#define DI2P_CHAIN_TN32 2
#endif
#ifndef DI2P_CHAIN_TN64
#define DI2P_CHAIN_TN64 1
#endif
#ifndef DI2P_CHAIN_W32
#define DI2P_CHAIN_W32 3          // resident workgroups per compute unit the chain kernels are launched with (<= what their registers allow)
#endif
#ifndef DI2P_CHAIN_W64
#define DI2P_CHAIN_W64 2
#endif
using Cfg128x128 = TileCfg<2, 2, 2, 2>;
using Cfg64x128 = TileCfg<2, 2, 1, 2>;
using Cfg32x128 = TileCfg<1, 4, 1, 1>;
using Cfg64x64 = TileCfg<2, 2, 1, 1>;

extern "C" int di2p_pointwise_gemm(const di2p_src_t* srcs, int n_src, const float* Wt, float* Y, int B, int M, int K, int N,
                                   const di2p_epilogue_t* epi, void* stream) {
    DI2P_CHECK_ARG(srcs && n_src >= 1 && n_src <= DI2P_MAX_SRC, "1..3 sources");
    DI2P_CHECK_ARG(B >= 0 && M >= 1 && K >= 1 && N >= 0, "bad size");
    DI2P_CHECK_ARG((long long)K * M < (1ll << 31), "weight too large");
    if (B == 0 || N == 0) return 0;
    SrcDev s{};
    int ctot = 0;
    for (int i = 0; i < DI2P_MAX_SRC; ++i) {
        if (i < n_src) {
            DI2P_CHECK_ARG(srcs[i].ptr && srcs[i].channels > 0, "bad source");
            DI2P_CHECK_ARG(srcs[i].mode != DI2P_SRC_GATHER || srcs[i].gidx, "gather source without index");
            DI2P_CHECK_ARG(srcs[i].mode != DI2P_SRC_GROUP || srcs[i].group >= 1, "group source without group");
            DI2P_CHECK_ARG((long long)srcs[i].channels * srcs[i].row_stride < (1ll << 31), "per-frame source extent must fit 31 bits");
            s.ptr[i] = srcs[i].ptr; s.gidx[i] = srcs[i].gidx; s.batch_stride[i] = srcs[i].batch_stride;
            s.row_stride[i] = srcs[i].row_stride; s.mode[i] = srcs[i].mode; s.group[i] = srcs[i].group > 0 ? srcs[i].group : 1;
            ctot += srcs[i].channels;
        }
        s.c_end[i] = ctot;
    }
    s.n_src = n_src;
    alias_absent_sources(s, n_src);
    DI2P_CHECK_ARG(ctot == K, "source channels do not sum to K");
    EpiDev e{};
    e.group_max = 1;
    if (epi) {
        e.scale = epi->scale; e.shift = epi->shift; e.batch_bias = epi->batch_bias; e.relu = epi->relu;
        e.group_max = epi->group_max > 1 ? epi->group_max : 1;
        for (int t = 0; t < 2; ++t) { e.g_table[t] = epi->g_table[t]; e.g_idx[t] = epi->g_idx[t]; e.g_w[t] = epi->g_w[t]; e.g_nodes[t] = epi->g_nodes[t]; }
        e.transpose_out = epi->transpose_out;
        e.gmax_out = epi->group_max > 1 ? epi->group_max_out : nullptr;
        e.gmax_dst = e.gmax_out ? e.gmax_out : Y;
        for (int t = 0; t < 2; ++t) {
            e.g_k[t] = e.g_table[t] ? epi->g_k[t] : 0;
            DI2P_CHECK_ARG(e.g_k[t] >= 0 && e.g_k[t] <= DI2P_MAX_GK, "g_k must be in [0, DI2P_MAX_GK]");
            DI2P_CHECK_ARG(!e.g_table[t] || (e.g_idx[t] && e.g_k[t] >= 1 && e.g_nodes[t] >= 1), "gathered table without index / k / nodes");
        }
        DI2P_CHECK_ARG(!(e.g_table[0] || e.g_table[1]) || (M % 4 == 0 && M >= 4), "gathered tables need M % 4 == 0");
        DI2P_CHECK_ARG(!e.transpose_out || (M % 4 == 0 && e.group_max == 1), "transpose_out needs M % 4 == 0 and no group_max");
        DI2P_CHECK_ARG(!epi->planes_out, "planes_out: only the bf16x3 entry points write split planes");
    }
    if (e.group_max > 1) {
        const int g = e.group_max;
        DI2P_CHECK_ARG((g & (g - 1)) == 0 && g <= 32 && N % g == 0, "group_max must be a power of two <= 32 dividing N");
    }
    hipStream_t st = (hipStream_t)stream;
    // 4-column staged path (weights 16-byte addressable, whole 4-column groups); sources that are all dense and 16-byte
    // addressable get one 16-byte load per row, gathered / group sources four dword loads
    // (narrow layers, M <= 64, are HBM/latency-bound and measured 25-35 % faster on the scalar stager below, whose K-step 16
    //  skips the rows k >= K instead of re-reading clamped ones)
    const bool vec = M > 64 && N % 4 == 0 && N >= 4 && M % 4 == 0 && aligned16(Wt) && !di2p_opt(DI2P_OPT_PW_NOVEC);
    bool dense = true;
    for (int i = 0; i < n_src; ++i)
        dense = dense && srcs[i].mode == DI2P_SRC_DENSE && srcs[i].row_stride % 4 == 0 && srcs[i].batch_stride % 4 == 0 && aligned16(srcs[i].ptr);
    if (vec) {
        // 64 x 64 tiles for every layer.  Alone, the big point layers are a few per cent faster on 128 x 128 tiles, but in the 8-stream
        // pipeline the small tile (32 KB of LDS and 100 registers per workgroup instead of 64 KB and 200) packs better beside the pose
        // solver's workgroups and the family's own serial time drops too (2.05 -> 1.95 ms): +1.5-2 % frames/s (tools/sweep_packing.sh).
        // DI2P_PW_CFG: 2 = 64x128, 3 = 128x128 tiles everywhere, 4 = 64x64 with K-step 16 (level with K-step 32); >= 16: 128x128 / 64x128 from that many workgroups on (the old rule: 1024)
        const long long wg128 = (long long)B * di2p_cdiv(N, 128) * di2p_cdiv(M, 128);
        const long long wg64x128 = (long long)B * di2p_cdiv(N, 128) * di2p_cdiv(M, 64);
        const long long opt = di2p_opt(DI2P_OPT_PW_CFG);
        const long long force = opt < 16 ? opt : 0, thr = opt >= 16 ? opt : (1ll << 62);
        if (force == 4 && N % 64 == 0) launch_pw_vec<TileCfg<2, 2, 1, 1, 16>>(dense, s, Wt, Y, B, M, K, N, e, st);
        else if (force == 3 || (!force && wg128 >= thr)) launch_pw_vec<TileCfg<2, 2, 2, 2, 32>>(dense, s, Wt, Y, B, M, K, N, e, st);
        else if (force == 2 || (!force && wg64x128 >= thr) || N % 64 != 0) launch_pw_vec<TileCfg<2, 2, 1, 2, 32>>(dense, s, Wt, Y, B, M, K, N, e, st);
        else launch_pw_vec<TileCfg<2, 2, 1, 1, 32>>(dense, s, Wt, Y, B, M, K, N, e, st);
        DI2P_RETURN_LAUNCH();
    }
    if (M <= 32) launch_pw<Cfg32x128>(s, Wt, Y, B, M, K, N, e, st);
    else if (M <= 64 || (long long)B * di2p_cdiv(N, 128) * di2p_cdiv(M, 128) < 256) launch_pw<Cfg64x128>(s, Wt, Y, B, M, K, N, e, st);
    else launch_pw<Cfg128x128>(s, Wt, Y, B, M, K, N, e, st);
    DI2P_RETURN_LAUNCH();
}

extern "C" long long di2p_bf16x3_packed_bytes(int K, int M) {
    if (K < 1 || M < 1) return 0;
    return (long long)(di2p_cdiv(K, X3_BK) * X3_BK / 8) * (di2p_cdiv(M, X3_BM) * X3_BM) * 48;
}

extern "C" int di2p_bf16x3_pack(const float* Wt, int K, int M, void* Wp, void* stream) {
    DI2P_CHECK_ARG(Wt && Wp && K >= 1 && M >= 1, "bad args");
    DI2P_CHECK_ARG(aligned16(Wp), "packed weights must be 16-byte aligned");
    const int Kp = di2p_cdiv(K, X3_BK) * X3_BK, Mp = di2p_cdiv(M, X3_BM) * X3_BM;
    hipLaunchKernelGGL(bf16x3_pack_kernel<0>, dim3(di2p_cdiv((long long)(Kp / 8) * Mp, 256)), dim3(256), 0, (hipStream_t)stream, Wt,
                       (unsigned short*)Wp, K, M, Kp, Mp, 0);
    DI2P_RETURN_LAUNCH();
}

// di2p_bf16x3_pack of the tap-major matrix of a 3 x 3 filter bank W f32[Cout][Cin][3][3], straight from W: dgrad == 0 the forward filter
// ([9 Cin, Cout]: Wp has di2p_bf16x3_packed_bytes(9 Cin, Cout) bytes), dgrad == 1 the filter of the input gradient ([9 Cout, Cin]: flipped
// taps, channel roles swapped).  The same bytes as permuting / flipping W on the host side and calling di2p_bf16x3_pack.
extern "C" int di2p_bf16x3_pack_conv3x3(const float* W, int Cout, int Cin, int dgrad, void* Wp, void* stream) {
    DI2P_CHECK_ARG(W && Wp && Cout >= 1 && Cin >= 1 && aligned16(Wp), "bad args");
    DI2P_CHECK_ARG((long long)Cout * Cin * 9 < (1ll << 31), "filter bank too large");
    const int ci_n = dgrad ? Cout : Cin, M = dgrad ? Cin : Cout, K = 9 * ci_n;
    const int Kp = di2p_cdiv(K, X3_BK) * X3_BK, Mp = di2p_cdiv(M, X3_BM) * X3_BM;
    if (ci_n % 8 == 0 && K == Kp) {          // whole k-groups per tap and no K padding (9 ci_n % 32 == 0 <=> ci_n % 32 == 0; else the general kernel)
        const dim3 grid(di2p_cdiv((long long)(ci_n / 8) * Mp, 256));
        if (dgrad) hipLaunchKernelGGL(bf16x3_pack_conv_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, W, (unsigned short*)Wp, M, Mp, ci_n);
        else hipLaunchKernelGGL(bf16x3_pack_conv_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, W, (unsigned short*)Wp, M, Mp, ci_n);
        DI2P_RETURN_LAUNCH();
    }
    const dim3 grid(di2p_cdiv((long long)(Kp / 8) * Mp, 256));
    if (dgrad) hipLaunchKernelGGL(bf16x3_pack_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, W, (unsigned short*)Wp, K, M, Kp, Mp, ci_n);
    else hipLaunchKernelGGL(bf16x3_pack_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, W, (unsigned short*)Wp, K, M, Kp, Mp, ci_n);
    DI2P_RETURN_LAUNCH();
}

// the epilogue of the two bf16x3 entry points (host struct -> device struct, argument checks)
static int x3_epilogue(const char* who, const di2p_epilogue_t* epi, float* Y, int M, int N, EpiDev& e) {
#define X3_EPI_CHECK(cond, msg) do { if (!(cond)) { di2p_set_error("%s: %s", who, msg); return -1; } } while (0)
    e = EpiDev{};
    e.group_max = 1;
    if (epi) {
        e.scale = epi->scale; e.shift = epi->shift; e.batch_bias = epi->batch_bias; e.relu = epi->relu;
        e.group_max = epi->group_max > 1 ? epi->group_max : 1;
        for (int t = 0; t < 2; ++t) { e.g_table[t] = epi->g_table[t]; e.g_idx[t] = epi->g_idx[t]; e.g_w[t] = epi->g_w[t]; e.g_nodes[t] = epi->g_nodes[t]; }
        e.transpose_out = epi->transpose_out;
        e.planes = (u32x2_t*)epi->planes_out;
        // the full-size output goes to Y, or to the planes; the group maxima to group_max_out when given, else to Y (which then holds nothing else)
        e.gmax_out = epi->group_max > 1 ? epi->group_max_out : nullptr;
        e.gmax_dst = e.gmax_out ? e.gmax_out : Y;
        for (int t = 0; t < 2; ++t) {
            e.g_k[t] = e.g_table[t] ? epi->g_k[t] : 0;
            X3_EPI_CHECK(e.g_k[t] >= 0 && e.g_k[t] <= DI2P_MAX_GK, "g_k must be in [0, DI2P_MAX_GK]");
            X3_EPI_CHECK(!e.g_table[t] || (e.g_idx[t] && e.g_k[t] >= 1 && e.g_nodes[t] >= 1), "gathered table without index / k / nodes");
        }
        X3_EPI_CHECK(!e.transpose_out || e.group_max == 1, "transpose_out excludes group_max");
        X3_EPI_CHECK(!e.planes || (!e.transpose_out && ((uintptr_t)e.planes & 15) == 0), "planes_out excludes transpose_out and must be 16-byte aligned");
        X3_EPI_CHECK(!e.planes || (long long)3 * M * N * 2 < (1ll << 31), "planes_out: a frame's planes must fit 31 bits");
    }
    X3_EPI_CHECK(Y || (e.planes && (e.group_max == 1 || e.gmax_out)), "Y may be NULL only when planes_out takes the full-size output");
    if (e.group_max > 1) {
        const int g = e.group_max;
        X3_EPI_CHECK((g & (g - 1)) == 0 && g <= 32 && N % g == 0, "group_max must be a power of two <= 32 dividing N");
    }
    return 0;
#undef X3_EPI_CHECK
}

// Same contract as di2p_pointwise_gemm with the weights given as di2p_bf16x3_pack's output (of the SAME [K][M] matrix).  Needs N % 4 == 0;
// every epilogue of the fp32 entry point is available, and planes_out.
extern "C" int di2p_pointwise_gemm_x3(const di2p_src_t* srcs, int n_src, const void* Wp, float* Y, int B, int M, int K, int N,
                                      const di2p_epilogue_t* epi, void* stream) {
    DI2P_CHECK_ARG(srcs && n_src >= 1 && n_src <= DI2P_MAX_SRC, "1..3 sources");
    DI2P_CHECK_ARG(Wp && aligned16(Wp), "null / misaligned pointer");
    DI2P_CHECK_ARG(B >= 0 && M >= 4 && M % 4 == 0 && K >= 1 && N >= 4 && N % 4 == 0, "needs M % 4 == 0 and N % 4 == 0");
    if (B == 0) return 0;
    SrcDev s{};
    int ctot = 0;
    for (int i = 0; i < DI2P_MAX_SRC; ++i) {
        if (i < n_src) {
            DI2P_CHECK_ARG(srcs[i].ptr && srcs[i].channels > 0, "bad source");
            DI2P_CHECK_ARG(srcs[i].mode != DI2P_SRC_GATHER || srcs[i].gidx, "gather source without index");
            DI2P_CHECK_ARG(srcs[i].mode != DI2P_SRC_GROUP || srcs[i].group >= 1, "group source without group");
            DI2P_CHECK_ARG((long long)srcs[i].channels * srcs[i].row_stride < (1ll << 31), "per-frame source extent must fit 31 bits");
            s.ptr[i] = srcs[i].ptr; s.gidx[i] = srcs[i].gidx; s.batch_stride[i] = srcs[i].batch_stride;
            s.row_stride[i] = srcs[i].row_stride; s.mode[i] = srcs[i].mode; s.group[i] = srcs[i].group > 0 ? srcs[i].group : 1;
            ctot += srcs[i].channels;
        }
        s.c_end[i] = ctot;
    }
    s.n_src = n_src;
    alias_absent_sources(s, n_src);
    DI2P_CHECK_ARG(ctot == K, "source channels do not sum to K");
    EpiDev e;
    if (x3_epilogue(__func__, epi, Y, M, N, e)) return -1;
    bool dense = true;
    for (int i = 0; i < n_src; ++i)
        dense = dense && srcs[i].mode == DI2P_SRC_DENSE && srcs[i].row_stride % 4 == 0 && srcs[i].batch_stride % 4 == 0 && aligned16(srcs[i].ptr);
    const int Mp = di2p_cdiv(M, X3_BM) * X3_BM;
    const dim3 grid(di2p_cdiv(N, X3_BN), di2p_cdiv(M, X3_BM), B);
    const hipStream_t st = (hipStream_t)stream;
    const u32x4_t* W = (const u32x4_t*)Wp;
    if (dense && e.planes) hipLaunchKernelGGL((pointwise_gemm_x3_kernel<true, true>), grid, dim3(256), 0, st, s, W, Y, M, K, N, Mp, e);
    else if (dense) hipLaunchKernelGGL((pointwise_gemm_x3_kernel<true, false>), grid, dim3(256), 0, st, s, W, Y, M, K, N, Mp, e);
    else if (e.planes) hipLaunchKernelGGL((pointwise_gemm_x3_kernel<false, true>), grid, dim3(256), 0, st, s, W, Y, M, K, N, Mp, e);
    else hipLaunchKernelGGL((pointwise_gemm_x3_kernel<false, false>), grid, dim3(256), 0, st, s, W, Y, M, K, N, Mp, e);
    DI2P_RETURN_LAUNCH();
}

extern "C" long long di2p_bf16x3_planes_bytes(int B, int C, int N) {
    if (B < 0 || C < 4 || C % 4 || N < 1) return 0;
    return (long long)B * 3 * C * N * 2;
}

// di2p_pointwise_gemm_x3 for ONE dense source given as the split planes a planes_out epilogue wrote (of a layer with M = this K rows and the
// same N): u16[B][3][K/4][N][4].  K % 32 == 0, N % 128 == 0; same epilogues; the same bits as the fp32-source entry point on the same values.
extern "C" int di2p_pointwise_gemm_x3p(const void* planes, const void* Wp, float* Y, int B, int M, int K, int N,
                                       const di2p_epilogue_t* epi, void* stream) {
    DI2P_CHECK_ARG(planes && Wp && aligned16(planes) && aligned16(Wp), "null / misaligned pointer");
    DI2P_CHECK_ARG(B >= 0 && M >= 4 && M % 4 == 0, "needs M % 4 == 0");
    DI2P_CHECK_ARG(K >= X3_BK && K % X3_BK == 0 && N >= X3_BN && N % X3_BN == 0, "planes source: K % 32 == 0 and N % 128 == 0");
    DI2P_CHECK_ARG((long long)3 * K * N * 2 < (1ll << 31), "a frame's planes must fit 31 bits");
    if (B == 0) return 0;
    EpiDev e;
    if (x3_epilogue(__func__, epi, Y, M, N, e)) return -1;
    const int Mp = di2p_cdiv(M, X3_BM) * X3_BM;
    const hipStream_t st = (hipStream_t)stream;
    if (M % 256 == 0 && di2p_opt(DI2P_OPT_PW_X3_PLANES) != 2) {          // 256-row tiles, both operands through LDS (knob value 2: the 128-row kernel)
        const dim3 grid(N / X3_BN, M / 256, B);
        auto k = e.planes ? pointwise_gemm_x3p8_kernel<true> : pointwise_gemm_x3p8_kernel<false>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, X3P8_LDS_BYTES);
        hipLaunchKernelGGL(k, grid, dim3(512), X3P8_LDS_BYTES, st, (const u32x4_t*)planes, (const u32x4_t*)Wp, Y, M, K, N, Mp, e);
        DI2P_RETURN_LAUNCH();
    }
    const dim3 grid(N / X3_BN, di2p_cdiv(M, X3_BM), B);
    if (e.planes) hipLaunchKernelGGL(pointwise_gemm_x3p_kernel<true>, grid, dim3(256), 0, st, (const u32x4_t*)planes, (const u32x4_t*)Wp, Y, M, K, N, Mp, e);
    else hipLaunchKernelGGL(pointwise_gemm_x3p_kernel<false>, grid, dim3(256), 0, st, (const u32x4_t*)planes, (const u32x4_t*)Wp, Y, M, K, N, Mp, e);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_point_head(const di2p_src_t* srcs, int n_src, const float* W0t, int K0, const di2p_epilogue_t* epi0,
                               const float* W1t, const float* scale1, const float* shift1, int relu1, const float* W2t,
                               const float* scale2, const float* shift2, int relu2, float* out, int B, int M, int P, int N,
                               void* stream) {
    DI2P_CHECK_ARG(srcs && n_src >= 1 && n_src <= DI2P_MAX_SRC && W0t && W1t && W2t && out && epi0, "null pointer");
    DI2P_CHECK_ARG(M == HEAD_M && P >= 1 && P <= 4, "fused head: hidden width 128, at most 4 outputs (use the separate layers otherwise)");
    DI2P_CHECK_ARG(B >= 0 && N >= 4 && N % 4 == 0 && K0 >= 1, "bad size");
    DI2P_CHECK_ARG(epi0->group_max <= 1 && !epi0->transpose_out && !epi0->planes_out, "fused head: layer 0 takes scale/shift/relu/bias/gathered only");
    if (B == 0) return 0;
    SrcDev s{};
    int ctot = 0;
    for (int i = 0; i < DI2P_MAX_SRC; ++i) {
        if (i < n_src) {
            DI2P_CHECK_ARG(srcs[i].ptr && srcs[i].channels > 0 && srcs[i].mode == DI2P_SRC_DENSE, "fused head: dense sources only");
            DI2P_CHECK_ARG(srcs[i].row_stride % 4 == 0 && srcs[i].batch_stride % 4 == 0 && aligned16(srcs[i].ptr), "sources must be 16-byte addressable");
            DI2P_CHECK_ARG((long long)srcs[i].channels * srcs[i].row_stride < (1ll << 31), "per-frame source extent must fit 31 bits");
            s.ptr[i] = srcs[i].ptr; s.batch_stride[i] = srcs[i].batch_stride; s.row_stride[i] = srcs[i].row_stride; s.mode[i] = DI2P_SRC_DENSE; s.group[i] = 1;
            ctot += srcs[i].channels;
        }
        s.c_end[i] = ctot;
    }
    s.n_src = n_src;
    alias_absent_sources(s, n_src);
    DI2P_CHECK_ARG(ctot == K0, "source channels do not sum to K0");
    DI2P_CHECK_ARG(aligned16(W0t) && aligned16(W1t), "weights must be 16-byte aligned");
    EpiDev e{};
    e.group_max = 1;
    e.scale = epi0->scale; e.shift = epi0->shift; e.batch_bias = epi0->batch_bias; e.relu = epi0->relu;
    for (int t = 0; t < 2; ++t) { e.g_table[t] = epi0->g_table[t]; e.g_idx[t] = epi0->g_idx[t]; e.g_w[t] = epi0->g_w[t]; e.g_nodes[t] = epi0->g_nodes[t]; }
    for (int t = 0; t < 2; ++t) {
        e.g_k[t] = e.g_table[t] ? epi0->g_k[t] : 0;
        DI2P_CHECK_ARG(e.g_k[t] >= 0 && e.g_k[t] <= DI2P_MAX_GK, "g_k must be in [0, DI2P_MAX_GK]");
        DI2P_CHECK_ARG(!e.g_table[t] || (e.g_idx[t] && e.g_k[t] >= 1 && e.g_nodes[t] >= 1), "gathered table without index / k / nodes");
    }
    HeadTail tl{W1t, scale1, shift1, W2t, scale2, shift2, relu1, relu2, P};
    bool reg_ok = K0 <= 96 && n_src <= 2 && di2p_opt(DI2P_OPT_HEAD_REG) != 0;
    if (n_src == 2) reg_ok = reg_ok && srcs[0].channels % 2 == 0;            // both half-waves of a K-step read the same source
    for (int i = 0; i < n_src; ++i) reg_ok = reg_ok && (long long)srcs[i].channels * srcs[i].row_stride * 4 < (1ll << 31);   // buffer-descriptor range
    if (reg_ok) {
        // wave-autonomous kernel: one 8-wave workgroup per compute unit (118 KB of weights in LDS), equal trips per wave where the sizes allow
        constexpr int KS0 = 48;
        const int nblk = di2p_cdiv(N, 32);
        const long long total = (long long)B * nblk;
        DI2P_CHECK_ARG(total < (1ll << 30), "too many column blocks");
        const int wgs = (int)std::min<long long>(di2p_cu_count(), (total + 7) / 8);
        const size_t lds_reg = (size_t)(2 * KS0 * HEAD_M + HEAD_M * HEAD_M + HEAD_M * 4 + 4 * HEAD_M) * sizeof(float);
        if (e.g_table[0] && e.g_table[1] && e.g_k[0] == 3 && e.g_k[1] == 3) {
            (void)hipFuncSetAttribute((const void*)point_head_reg_kernel<true, KS0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reg);
            hipLaunchKernelGGL((point_head_reg_kernel<true, KS0>), dim3(wgs), dim3(512), lds_reg, (hipStream_t)stream, s, W0t, K0, e, tl, out, N, nblk, (int)total);
        } else {
            (void)hipFuncSetAttribute((const void*)point_head_reg_kernel<false, KS0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reg);
            hipLaunchKernelGGL((point_head_reg_kernel<false, KS0>), dim3(wgs), dim3(512), lds_reg, (hipStream_t)stream, s, W0t, K0, e, tl, out, N, nblk, (int)total);
        }
        DI2P_RETURN_LAUNCH();
    }
    const size_t lds = (HeadCfg::LDS_FLOATS + HEAD_M * HEAD_BN) * sizeof(float);
    // > 64 KB of dynamic LDS needs the opt-in (per device; the call is cheap, so it is simply made every time)
    if (e.g_table[0] && e.g_table[1] && e.g_k[0] == 3 && e.g_k[1] == 3) {      // the reference's configuration (k_interp_point_a = k_interp_point_b = 3)
        (void)hipFuncSetAttribute((const void*)point_head_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(point_head_kernel<true>, dim3(di2p_cdiv(N, HEAD_BN), B), dim3(HeadCfg::THREADS), lds, (hipStream_t)stream, s, W0t, K0, e, tl, out, N);
    } else {
        (void)hipFuncSetAttribute((const void*)point_head_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(point_head_kernel<false>, dim3(di2p_cdiv(N, HEAD_BN), B), dim3(HeadCfg::THREADS), lds, (hipStream_t)stream, s, W0t, K0, e, tl, out, N);
    }
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_point_chain(const di2p_src_t* srcs, int n_src, const float* W0t, int K0, const di2p_epilogue_t* epi0,
                                const float* W1t, const float* scale1, const float* shift1, int relu1, const float* W2t,
                                const float* scale2, const float* shift2, int relu2, float* Y, int B, int M, int N, void* stream) {
    DI2P_CHECK_ARG(srcs && n_src == 1 && W0t && W1t && epi0, "fused chain: one dense source");
    DI2P_CHECK_ARG(M == 32 || M == 64, "fused chain: width 32 or 64 (use the separate layers otherwise)");
    DI2P_CHECK_ARG(K0 <= M, "fused chain: at most M input channels");
    DI2P_CHECK_ARG(B >= 0 && N >= 1 && K0 >= 1, "bad size");
    DI2P_CHECK_ARG(epi0->group_max <= 1 && !epi0->transpose_out && !epi0->planes_out, "fused chain: layer 0 takes scale/shift/relu/bias/gathered only");
    if (B == 0) return 0;          // an empty batch has no output buffer to check
    DI2P_CHECK_ARG(Y && srcs[0].ptr && srcs[0].channels == K0 && srcs[0].mode == DI2P_SRC_DENSE, "fused chain: one dense source of K0 channels");
    DI2P_CHECK_ARG((long long)srcs[0].channels * srcs[0].row_stride < (1ll << 31), "per-frame source extent must fit 31 bits");
    EpiDev e{};
    e.group_max = 1;
    e.scale = epi0->scale; e.shift = epi0->shift; e.batch_bias = epi0->batch_bias; e.relu = epi0->relu;
    for (int t = 0; t < 2; ++t) { e.g_table[t] = epi0->g_table[t]; e.g_idx[t] = epi0->g_idx[t]; e.g_w[t] = epi0->g_w[t]; e.g_nodes[t] = epi0->g_nodes[t]; }
    for (int t = 0; t < 2; ++t) {
        e.g_k[t] = e.g_table[t] ? epi0->g_k[t] : 0;
        DI2P_CHECK_ARG(e.g_k[t] >= 0 && e.g_k[t] <= DI2P_MAX_GK, "g_k must be in [0, DI2P_MAX_GK]");
        DI2P_CHECK_ARG(!e.g_table[t] || (e.g_idx[t] && e.g_k[t] >= 1 && e.g_nodes[t] >= 1), "gathered table without index / k / nodes");
    }
    ChainTail tl{W1t, scale1, shift1, W2t, scale2, shift2, relu1, relu2};
    const float* X = srcs[0].ptr;
    const long long x_bs = srcs[0].batch_stride;
    const int x_rs = (int)srcs[0].row_stride;
    hipStream_t st = (hipStream_t)stream;
    // Grid: whole workgroups per compute unit (w = waves per SIMD the kernel's registers allow) and as few, equal trips per wave as possible
    const int cus = di2p_cu_count();
#define DI2P_CHAIN_LAUNCH(MM, KS0, NL, TN, WMAX)                                                                                    \
    do {                                                                                                                            \
        const int nblk = di2p_cdiv(N, 32 * (TN));                                                                                   \
        const long long total = (long long)B * nblk;                                                                                \
        DI2P_CHECK_ARG(total < (1ll << 30), "too many column blocks");                                                              \
        int best_w = 1;                                                                                                             \
        long long best_trips = 1ll << 62;                                                                                           \
        for (int w = (WMAX); w >= 1; --w) {                                                                                         \
            const long long trips = (total + 4ll * cus * w - 1) / (4ll * cus * w);                                                  \
            if (trips < best_trips) { best_trips = trips; best_w = w; }                                                             \
        }                                                                                                                           \
        const int wgs = (int)std::min<long long>((long long)cus * best_w, (total + 3) / 4);                                         \
        const size_t lds = (size_t)(2 * (KS0) * (MM) + ((NL) - 1) * (MM) * (MM) + 6 * (MM)) * sizeof(float);                        \
        hipLaunchKernelGGL((point_chain_kernel<MM, KS0, NL, TN>), dim3(wgs), dim3(256), lds, st, X, x_bs, x_rs, W0t, K0, e, tl, Y, N, \
                           nblk, (int)total);                                                                                       \
    } while (0)
    const bool three = W2t != nullptr;
    if (M == 32) {
        if (K0 <= 8) { if (three) DI2P_CHAIN_LAUNCH(32, 4, 3, DI2P_CHAIN_TN32, DI2P_CHAIN_W32); else DI2P_CHAIN_LAUNCH(32, 4, 2, DI2P_CHAIN_TN32, DI2P_CHAIN_W32); }
        else           { if (three) DI2P_CHAIN_LAUNCH(32, 16, 3, DI2P_CHAIN_TN32, DI2P_CHAIN_W32); else DI2P_CHAIN_LAUNCH(32, 16, 2, DI2P_CHAIN_TN32, DI2P_CHAIN_W32); }
    } else {
        if (K0 <= 32) { if (three) DI2P_CHAIN_LAUNCH(64, 16, 3, DI2P_CHAIN_TN64, DI2P_CHAIN_W64); else DI2P_CHAIN_LAUNCH(64, 16, 2, DI2P_CHAIN_TN64, DI2P_CHAIN_W64); }
        else           { if (three) DI2P_CHAIN_LAUNCH(64, 32, 3, DI2P_CHAIN_TN64, DI2P_CHAIN_W64); else DI2P_CHAIN_LAUNCH(64, 32, 2, DI2P_CHAIN_TN64, DI2P_CHAIN_W64); }
    }
#undef DI2P_CHAIN_LAUNCH
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_batch_gemv(const float* Wt, int M, int k0, const float* v, int Kv, float* out, int B, void* stream) {
    DI2P_CHECK_ARG(Wt && v && out && M >= 1 && Kv >= 1 && k0 >= 0 && B >= 0, "bad args");
    if (B == 0) return 0;
    hipLaunchKernelGGL(batch_gemv_kernel, dim3(di2p_cdiv(M, 64), B), dim3(256), 0, (hipStream_t)stream, Wt, M, k0, v, Kv, 0,
                       (const float*)nullptr, 0, out);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_batch_gemv2(const float* Wt, int M, int k0, const float* v0, int Kv0, int k1, const float* v1, int Kv1,
                                float* out, int B, void* stream) {
    DI2P_CHECK_ARG(Wt && v0 && v1 && out && M >= 1 && Kv0 >= 1 && Kv1 >= 1 && k0 >= 0 && k1 >= 0 && B >= 0, "bad args");
    if (B == 0) return 0;
    hipLaunchKernelGGL(batch_gemv_kernel, dim3(di2p_cdiv(M, 64), B), dim3(256), 0, (hipStream_t)stream, Wt, M, k0, v0, Kv0, k1, v1, Kv1, out);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_attention_pool(const float* feat, const float* score, float* out, int B, int C, int HW, int Mn, void* stream) {
    DI2P_CHECK_ARG(feat && score && out && B >= 0 && C >= 1 && HW >= 1 && Mn >= 1, "bad args");
    if (B == 0) return 0;
    using Cfg = Cfg64x64;
    hipLaunchKernelGGL(attention_pool_kernel<Cfg>, dim3(di2p_cdiv(Mn, Cfg::BN), di2p_cdiv(C, Cfg::BM), B), dim3(Cfg::THREADS),
                       Cfg::LDS_FLOATS * sizeof(float), (hipStream_t)stream, feat, score, out, C, HW, Mn);
    DI2P_RETURN_LAUNCH();
}
