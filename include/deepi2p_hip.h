/*
 * deepi2p_hip.h -- C ABI of libdeepi2p_hip.so: the MI355X (gfx950) implementation of the
 * DeepI2P registration hot path.  Plain pointers and sizes only; every pointer is a DEVICE
 * pointer (HBM) unless the name ends in _host; `stream` is a hipStream_t passed as void*
 * (NULL = the null stream).  No function allocates, frees or synchronises: all are safe to
 * capture into a hipGraph.  Return value: 0 on success, otherwise a hipError_t / negative
 * argument-check code; di2p_last_error() returns a static description.
 *
 * Each entry point names the reference interface (path:line under lijx10/DeepI2P) it replaces.
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Tensor layouts are the reference's own: point features f32 [B,C,N] (N contiguous), images
 * f32 NCHW, indices i32.
 */
#ifndef DEEPI2P_HIP_H
#define DEEPI2P_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* di2p_last_error(void);
int di2p_version(void);
/* Tuning / test knobs, cached in the library (initialised once from the environment variable DI2P_<NAME>): "conv_nosplit",
 * "conv_split_blocks", "conv_novec", "conv_cfg", "conv_depth1", "conv_x3", "conv_x3_cfg", "head_x3", "head_x3_tab", "stem_x3", "pw_x3", "pw_novec", "solver_cfg", "solver_nocull", "solver_noprefilter", "solver_tier_sweeps".
 * set: 0, or -1 for an unknown name; get: the value, or -1 for an unknown name. */
int di2p_set_option(const char* name, long long value);
long long di2p_get_option(const char* name);

/* ---------------------------------------------------------------------------------------------
 * index_max  -- replaces models/index_max_ext: index_max.cpp:141-148 (forward_cuda_shared_mem),
 * :132-139 (forward_cuda); kernels index_max_cuda.cu:10-26, :30-62.
 *   data f32[B,C,N], index i32[B,N] with values in [0,K)  ->  max_idx i32[B,C,K]
 *   max_idx[b,c,k] = first n with index[b,n]==k attaining max(data[b,c,n]) ; floor -1000 ;
 *   empty cluster (or nothing > -1000) -> 0.
 * di2p_index_max_values additionally writes the maxima themselves with empty clusters zeroed,
 * i.e. data.gather(2,max_idx) * mask_row_max of models/networks_pc.py:91-92,103-104 (max_idx may
 * be NULL there; mask f32[B,K] = mask_row_max from di2p_cluster_stats, or NULL = "a node is
 * non-empty iff something beat the floor").  `workspace` must hold B*C*K uint64 (scratch, contents undefined after). */
int di2p_index_max_forward(const float* data, const int32_t* index, int32_t* max_idx,
                           int B, int C, int N, int K, void* workspace, void* stream);
int di2p_index_max_values(const float* data, const int32_t* index, const float* mask, float* max_val,
                          int32_t* max_idx, int B, int C, int N, int K, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ball_query -- replaces models/ball_query_ext: ball_query.cpp:33-39 (forward_cuda_shared_mem),
 * kernel ball_query_cuda.cu:11-50.  node_to_point_dist f32[B,M,N] -> i32[B,M,K]: first K point
 * ids (ascending n) with dist <= radius, cyclically padded; all zero when none. */
int di2p_ball_query_forward(const float* node_to_point_dist, int32_t* out_idx, float radius, int K,
                            int B, int M, int N, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Point <-> node kernels of PCEncoder / KeypointDetector.
 *
 * di2p_knn_nodes: the dense-distance + torch.topk(k smallest, sorted) idiom of
 *   models/networks_pc.py:61-65, models/networks_united.py:158-161,176-178,
 *   models/layers_pc.py:798-799.   query f32[B,3,Nq], nodes f32[B,3,M] ->
 *   idx i32[B,Nq,k] ascending distance, ties -> lower node id.  weights (may be NULL) f32[B,Nq,k]
 *   = 1 - d_k/sum_j d_j, the interpolation weights of networks_united.py:97-98.
 *   k <= 16, M <= 1024.
 * di2p_cluster_stats: networks_pc.py:66-76.  pc f32[B,3,N], knn idx (nearest = idx[b,n,0], row
 *   stride idx_stride) -> cluster_mean f32[B,3,M] = sum/(count+1e-5), mask f32[B,M] (1 if the
 *   node owns >=1 point), min_idx i32[B,N] (compact copy of the nearest id).
 * di2p_build_point_input: networks_pc.py:79-85.  -> pc_centers f32[B,3,N],
 *   augmented f32[B,7,N] = cat(pc - centers, intensity, sn).
 * di2p_interpolate: networks_united.py:90-103 (upsample_by_interpolation given idx+weights):
 *   feats f32[B,C,M] -> out f32[B,C,Nq] = sum_k w[b,n,k] * feats[b,c,idx[b,n,k]].
 * di2p_gather_neighbors: layers_pc.py:800-807 coordinates part: out f32[B,3,Mq*K] =
 *   database[:, idx] - query.
 * di2p_argmax_channels: multimodal_classifier.py:115-116.  scores f32[B,C,N] -> i32[B,N], first
 *   maximum wins. */
int di2p_knn_nodes(const float* query, const float* nodes, int32_t* idx, float* weights,
                   int B, int Nq, int M, int k, void* stream);
int di2p_cluster_stats(const float* pc, const int32_t* knn_idx, int idx_stride, float* cluster_mean,
                       float* mask, int32_t* min_idx, int B, int N, int M, void* stream);
int di2p_build_point_input(const float* pc, const float* intensity, const float* sn,
                           const float* cluster_mean, const int32_t* min_idx, float* pc_centers,
                           float* augmented, int B, int N, int M, void* stream);
int di2p_interpolate(const float* feats, const int32_t* idx, const float* weights, float* out,
                     int B, int C, int M, int Nq, int k, void* stream);
int di2p_gather_neighbors(const float* database, const float* query, const int32_t* idx, float* out,
                          int B, int Md, int Mq, int K, void* stream);
int di2p_argmax_channels(const float* scores, int32_t* out, int B, int C, int N, long long batch_stride /* elements; 0 = C*N */, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pointwise contraction = EquivariantLayer / MyConv2d(1x1) (+BN eval +ReLU) of
 * models/layers_pc.py:259-342, :110-190, with the torch.cat / expand / gather feeding it
 * (networks_pc.py:98, layers_pc.py:808-813, networks_united.py:139-197) folded into the operand
 * loader.  fp32 in, fp32 MFMA (v_mfma_f32_32x32x2_f32) accumulate, fp32 out.
 *
 *   Y[b,m,n'] = epi( sum_k Wt[k,m] * X[b,k,n] )        n' = n, or n / group_max when group_max > 1
 *   epi(a)    = relu?( scale[m]*(a + batch_bias[b,m] + sum_t sum_j gw_t[b,n,j]*G_t[b,m,gidx_t[b,n,j]]) + shift[m] )
 *   (scale/shift = folded BatchNorm(eval) + conv bias; the batch_bias and gathered terms are part of
 *   the pre-activation, i.e. they stand for input channels that were never materialised)
 *
 * X is the virtual concatenation along k of up to DI2P_MAX_SRC sources.  Wt is the layer weight
 * transposed to [K,M] (packed once at load time).  group_max > 1 takes the max over each run of
 * `group_max` consecutive columns (torch.max(dim=3) of layers_pc.py:811,816). */
#define DI2P_MAX_SRC 3
#define DI2P_MAX_GK 16      /* = the k limit of di2p_knn_nodes */
enum { DI2P_SRC_DENSE = 0,   /* X[b,c,n]      = ptr[b*batch_stride + c*row_stride + n]            */
       DI2P_SRC_GATHER = 1,  /* X[b,c,n]      = ptr[b*batch_stride + c*row_stride + gidx[b,n]]    */
       DI2P_SRC_GROUP = 2    /* X[b,c,n]      = ptr[b*batch_stride + c*row_stride + n / group]    */ };
typedef struct {
    const float* ptr;
    const int32_t* gidx;     /* GATHER: i32[B, N] (batch stride = N) */
    long long batch_stride;  /* elements */
    int row_stride;          /* elements */
    int channels;
    int mode;
    int group;               /* GROUP mode divisor */
    int pad_;
} di2p_src_t;

typedef struct {
    const float* scale;        /* [M] or NULL (=1) */
    const float* shift;        /* [M] or NULL (=0) */
    const float* batch_bias;   /* [B,M] or NULL    */
    int relu;
    int group_max;             /* 1 = none */
    /* optional gathered add (per_point_pn layer 0 with W*interp == interp(W*nodes)):            */
    const float* g_table[2];   /* each f32[B,g_nodes,M] (node-major, M % 4 == 0; what transpose_out writes) or NULL */
    const int32_t* g_idx[2];   /* i32[B,N,g_k[t]] */
    const float* g_w[2];       /* f32[B,N,g_k[t]], or NULL for unit weights (plain gather) */
    int g_nodes[2];
    int g_k[2];                /* neighbours per column of each table, 1..DI2P_MAX_GK (the two tables may differ:
                                  opt.k_interp_point_a / k_interp_point_b, networks_united.py:158-165,188-191) */
    int transpose_out;         /* 1: Y is written f32[B,N,M] (needs M % 4 == 0, group_max == 1) */
    float* group_max_out;      /* with group_max > 1: NULL -> Y holds the maxima [B,M,N/group_max]; else Y is written in full
                                  [B,M,N] and the maxima go here (layers_pc.py:809-813 needs both) */
    void* planes_out;          /* (ABI 6; the bf16x3 entry points only, NULL elsewhere) not NULL: the full-size output [B,M,N] is written
                                  HERE, already split for the layer that contracts over it, instead of to Y as fp32:
                                  u16[B][3][M/4][N][4] = three bf16 planes (x = x1 + x2 + x3, exact), four consecutive rows of one
                                  column per 8 bytes; di2p_bf16x3_planes_bytes(B, M, N) bytes, 16-byte aligned, M % 4 == 0, no
                                  transpose_out.  Y then only receives the group maxima (group_max > 1 and group_max_out == NULL) and
                                  may be NULL otherwise. */
} di2p_epilogue_t;

int di2p_pointwise_gemm(const di2p_src_t* srcs_host, int n_src, const float* Wt, float* Y,
                        int B, int M, int K, int N, const di2p_epilogue_t* epi_host, void* stream);
/* The same contraction on the bf16 matrix instructions with an EXACT three-way split of both fp32 operands (x = x1 + x2 + x3, three truncated
 * bf16 terms; six bf16 products per fp32 product, fp32 accumulation): as accurate against an fp64 contraction as the fp32-MFMA kernel, 2.67x
 * its matrix-pipe rate.  For the GEMM-shaped layers (Conv1d / Conv2d(1x1) of models/layers_pc.py:259-408,779-818 with K >= 128 input
 * channels); the weights are split once:  Wp = di2p_bf16x3_pack(Wt [K][M])  (di2p_bf16x3_packed_bytes(K, M) bytes, 16-byte aligned).
 * Same sources, epilogues and output layouts as di2p_pointwise_gemm; needs M % 4 == 0 and N % 4 == 0. */
long long di2p_bf16x3_packed_bytes(int K, int M);
int di2p_bf16x3_pack(const float* Wt, int K, int M, void* Wp, void* stream);
/* (ABI 6, training) the split tap-major matrix of a 3x3 filter bank W f32[Cout][Cin][3][3] for di2p_conv3x3_x3, straight from W:
 * dgrad == 0: of the forward filter ([9 Cin, Cout]); dgrad == 1: of the filter that computes the layer's INPUT gradient as a convolution of
 * dY ([9 Cout, Cin]: taps flipped, channel roles swapped; models/resnet.py:56-72 under autograd).  Same bytes as di2p_bf16x3_pack of that matrix. */
int di2p_bf16x3_pack_conv3x3(const float* W, int Cout, int Cin, int dgrad, void* Wp, void* stream);
int di2p_pointwise_gemm_x3(const di2p_src_t* srcs_host, int n_src, const void* Wp, float* Y,
                           int B, int M, int K, int N, const di2p_epilogue_t* epi_host, void* stream);
/* A chain of such layers (GeneralKNNFusionModule's layers_before.1 -> layers_after.0 -> layers_after.1, models/layers_pc.py:779-818) hands
 * its activations on ALREADY SPLIT: the producer's epilogue writes planes_out (6 bytes per value instead of 4), the consumer stages them into
 * LDS without arithmetic -- the split leaves the K loop and is done once per value instead of once per 128-row block of every consumer.
 * `planes` = a planes_out of a layer with K rows and the same N; K % 32 == 0, N % 128 == 0.  Same epilogues (planes_out included) and, on the
 * same values, the same bits as di2p_pointwise_gemm_x3 with that layer's fp32 output as its one dense source. */
long long di2p_bf16x3_planes_bytes(int B, int C, int N);
int di2p_pointwise_gemm_x3p(const void* planes, const void* Wp, float* Y,
                            int B, int M, int K, int N, const di2p_epilogue_t* epi_host, void* stream);

/* Y[b,m] = sum_k Wt[k0+k, m] * v[b,k]  (the broadcast part of a concatenated input, folded into
 * batch_bias: networks_united.py:139-155,170-187 expand()s).  v f32[B,Kv]. */
/* Fused per-point head (per_point_pn, networks_united.py:57-74,194-197, coarse variant): three pointwise layers in
 * one launch, out = L2(L1(L0(concat(srcs)))) with
 *   L0: K0 dense channels -> M = 128, epilogue epi0 (scale/shift/relu, batch_bias, gathered node tables),
 *   L1: 128 -> 128 (W1t f32[128,128] k-major, scale1/shift1, relu1),   L2: 128 -> P <= 4 (W2t f32[128,P], scale2/shift2 or NULL).
 * out f32[B,P,N].  Bit-identical to three di2p_pointwise_gemm calls; the hidden activations stay in LDS. */
int di2p_point_head(const di2p_src_t* srcs_host, int n_src, const float* W0t, int K0, const di2p_epilogue_t* epi0_host,
                    const float* W1t, const float* scale1, const float* shift1, int relu1,
                    const float* W2t, const float* scale2, const float* shift2, int relu2,
                    float* out, int B, int M, int P, int N, void* stream);

/* Fused narrow PointNet chain (first_pointnet / second_pointnet of PCEncoder, networks_pc.py:36-43,60-75): two or three pointwise
 * layers of one width M (32 or 64) in one launch, Y = L2(L1(L0(src))) or L1(L0(src)) when W2t == NULL, with
 *   L0: ONE dense source of K0 <= M channels -> M, epilogue epi0 (scale/shift/relu, batch_bias, gathered node tables);
 *   L1, L2: M -> M (W1t / W2t f32[M,M] k-major, scale/shift may be NULL).  Y f32[B,M,N].
 * Bit-identical to the separate di2p_pointwise_gemm calls (up to the sign of zero); the hidden activations stay in registers. */
int di2p_point_chain(const di2p_src_t* srcs_host, int n_src, const float* W0t, int K0, const di2p_epilogue_t* epi0_host,
                     const float* W1t, const float* scale1, const float* shift1, int relu1, const float* W2t,
                     const float* scale2, const float* shift2, int relu2, float* Y, int B, int M, int N, void* stream);

int di2p_batch_gemv(const float* Wt, int M, int k0, const float* v, int Kv, float* out, int B, void* stream);
/* Two broadcast inputs in one launch: Y[b,m] = (sum_k Wt[k0+k,m] v0[b,k]) + (sum_k Wt[k1+k,m] v1[b,k]), each sum formed
 * exactly as di2p_batch_gemv forms it (node_b_pn: global PC feature + global image feature, networks_united.py:152-155). */
int di2p_batch_gemv2(const float* Wt, int M, int k0, const float* v0, int Kv0, int k1, const float* v1, int Kv1,
                     float* out, int B, void* stream);

/* Batched attention contraction of networks_united.py:147-150,170-174:
 * out[b,c,m] = (1/HW) * sum_hw feat[b,c,hw] * score[b,hw,m]. */
int di2p_attention_pool(const float* feat, const float* score, float* out, int B, int C, int HW, int Mn, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Image branch: ResNet-34 of models/resnet.py:56-72,125-216 (BasicBlock conv-bn-relu, residual
 * add, 7x7/2 stem, 3x3/2 max-pool, global average pool) as implicit-GEMM convolutions on fp32
 * MFMA with the BN(eval)+residual+ReLU epilogue fused.
 *   x f32[B,Cin,H,W] (NCHW), Wt f32[Cin*KH*KW, Cout] (weight[co,ci,kh,kw] transposed), y
 *   f32[B,Cout,OH,OW];  y = relu?( scale*conv(x) + shift + residual ).
 *   tap_major != 0: Wt rows are ordered (kh,kw,ci) instead of the weight's own (ci,kh,kw); needs Cin % 16 == 0 and
 *   lets the loader decode the filter tap once per 16-row K-step. */
int di2p_conv2d(const float* x, const float* Wt, const float* scale, const float* shift,
                const float* residual, float* y, int B, int Cin, int H, int W, int Cout,
                int KH, int KW, int stride, int pad, int relu, int tap_major, void* stream);
/* Same, with scratch for split-K: layers with few output pixels (ResNet stage 4) are cut into K-slices along the
 * filter taps, the partial sums (slices x B*Cout*OH*OW floats) are combined in slice order by a second pass.
 * di2p_conv2d_workspace_bytes returns the size this shape wants (0: no split); a NULL / too small workspace
 * silently runs unsplit.  Results are deterministic either way. */
int di2p_conv2d_ws(const float* x, const float* Wt, const float* scale, const float* shift,
                   const float* residual, float* y, int B, int Cin, int H, int W, int Cout,
                   int KH, int KW, int stride, int pad, int relu, int tap_major,
                   void* workspace, long long workspace_bytes, void* stream);
long long di2p_conv2d_workspace_bytes(int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                                      int tap_major);
/* 3x3 / stride 1 / pad 1 convolutions (26 of the 36 of ResNet-34, models/resnet.py:56-72) as a fused Winograd F(2x2,3x3) kernel:
 * 16 multiplications per 2x2 output tile and (ci, co) pair instead of 36; exact in exact arithmetic.
 *   di2p_winograd_weight_transform: weight f32[Cout,Cin,3,3] -> U f32[16,Cin,Cout] (= G g G^T), once per checkpoint load.
 *   di2p_conv3x3_winograd: y = relu?( scale * conv(x) + shift + residual ); needs Cin % 8 == 0 and Cout % 32 == 0. */
int di2p_winograd_weight_transform(const float* weight, float* U, int Cin, int Cout, void* stream);
/* round 6 (training path): the transformed filter of the INPUT GRADIENT of a stride-1 3x3 layer, U f32[16,Cin_g,Cout_g], straight from the layer's
 * forward filter weight f32[Cout_f = Cin_g, Cin_f = Cout_g, 3, 3] (taps flipped, channel roles swapped).  Replaces torch's flip + transpose + copy
 * in front of di2p_winograd_weight_transform (models/multimodal_classifier.py:213-218 reaches it through loss.backward()). */
int di2p_winograd_weight_transform_dgrad(const float* weight, float* U, int Cin_g, int Cout_g, void* stream);
int di2p_conv3x3_winograd(const float* x, const float* U, const float* scale, const float* shift, const float* residual, float* y, int B,
                          int Cin, int H, int W, int Cout, int relu, void* stream);
/* The coarse per-point head (per_point_pn, models/networks_united.py:57-74 applied at :188-197: 736 -> 128 -> 128 -> P) in ONE launch on the bf16
 * matrix instructions with exact three-way fp32 splits, wave-autonomous (a wave keeps all 128 channels of 32 points in registers through the three
 * layers).  Layer 0 = the dense channels of the point (two sources f32[B, ch, N], channels multiples of 16) through W0 PLUS the per-node products
 * of its interpolated inputs, gathered from two node-major tables f32[B, nodes, 128] (3 neighbours each, idx i32[B,N,3], weights f32[B,N,3] or NULL
 * = 1) -- the same decomposition as di2p_point_head.  Weights: di2p_head_x3_pack of the [K][128] k-major matrices of layer 0 (rows of the dense
 * sources only) and layer 1; scale_shift = f32[4][128] (scale0, shift0, scale1, shift1); W2t f32[128][P], scale2 / shift2 f32[P] or NULL.
 * out f32[B,P,N].  This build runs 32 + 64 dense channels (K0 = 96), P <= 4; anything else: di2p_point_head / di2p_pointwise_gemm. */
typedef struct {
    const float* src[2]; long long batch_stride[2]; int row_stride[2]; int channels[2];
    const void* W0p; const void* W1p; const float* scale_shift; int relu0, relu1;
    const float* tab[2]; const int32_t* idx[2]; const float* w[2]; int nodes[2];
    const float* W2t; const float* scale2; const float* shift2; int relu2, P;
} di2p_head_x3_t;
long long di2p_head_x3_packed_bytes(int K);
int di2p_head_x3_pack(const float* Wt, int K, void* Wp, void* stream);
int di2p_point_head_x3(const di2p_head_x3_t* h, float* out, int B, int N, void* stream);
/* 3x3 convolutions (pad 1, stride 1 or 2) on the bf16 matrix instructions with the EXACT three-way fp32 split of both operands ("bf16x3",
 * see di2p_pointwise_gemm_x3): a direct implicit GEMM whose input patch is split once while it is staged into LDS and then serves all nine
 * taps.  Replaces cuDNN's conv+BN+ReLU(+residual) of models/resnet.py:56-72 (BasicBlock.forward) for the layers it supports; with stride 2
 * the BasicBlock's 1x1 / stride-2 downsample branch (models/resnet.py:160-164, applied at :62-63) is computed from the same staged patch.
 *   Wp    = di2p_bf16x3_pack of the tap-major matrix Wt[(kh*3+kw)*Cin + ci][Cout] (K = 9 Cin, M = Cout);
 *   Wp_ds = di2p_bf16x3_pack of Wt_ds[Cin][Cout] (stride 2: required; stride 1: must be NULL);
 *   y     f32[B,Cout,OH,OW] = relu?(scale * conv3x3(x f32[B,Cin,H,W]) + shift + residual);  y_ds = scale_ds * conv1x1/s2(x) + shift_ds.
 * di2p_conv3x3_x3_supported: 1 if a kernel instance exists for the shape (OW % 32 == 0 and Cin % 16 == 0, or OW % 16 == 0 and
 * Cin % 32 == 0; the input patch of a tile must fit the LDS), else 0 -- the caller then uses di2p_conv3x3_winograd / di2p_conv2d. */
int di2p_conv3x3_x3_supported(int B, int Cin, int H, int W, int Cout, int stride);
int di2p_conv3x3_x3(const float* x, const void* Wp, const float* scale, const float* shift, const float* residual, float* y, int B,
                    int Cin, int H, int W, int Cout, int stride, int relu, const void* Wp_ds, const float* scale_ds,
                    const float* shift_ds, float* y_ds, void* stream);
/* The ResNet stem (7x7 / stride 2 / pad 3, 3 -> 64 channels, models/resnet.py:137-139,197-199) as a direct kernel: the filter bank and
 * the input row segments of a workgroup are staged once (columns de-interleaved by parity), then 168 MFMAs per wave without a barrier.
 *   di2p_stem_pack: weight f32[64,3,7,7] -> Wp f32[168,64] (taps padded 7 -> 8 per row), once per checkpoint load.
 *   di2p_conv7x7s2_stem: y f32[B,64,OH,OW] = relu?(scale * conv(x f32[B,3,H,W]) + shift). */
int di2p_stem_pack(const float* weight, float* Wp, void* stream);
int di2p_conv7x7s2_stem(const float* x, const float* Wp, const float* scale, const float* shift, float* y, int B, int H, int W, int relu,
                        void* stream);
int di2p_maxpool3x3s2(const float* x, float* y, int B, int C, int H, int W, void* stream);
/* The same head of the image branch -- conv1 7x7/2 + bn1 + relu + maxpool 3x3/2 (models/resnet.py:137-141,197-201) -- as ONE launch on the
 * bf16 matrix instructions with exact three-way fp32 splits (six bf16 products per fp32 product, fp32 accumulation: as accurate against fp64
 * as the fp32-MFMA stem); the 64 x H/2 x W/2 activation between the convolution and the pool stays in the compute unit.
 *   di2p_stem_x3_pack: weight f32[64,3,7,7] -> Wp (di2p_stem_x3_packed_bytes() bytes, 16-byte aligned), once per checkpoint load.
 *   di2p_stem_x3_supported: 1 if the image size runs (H % 4 == 0, W % 128 == 0, W <= 512), else 0 -- the caller then uses the two launches above.
 *   di2p_stem_x3: y f32[B,64,H/4,W/4] = maxpool3x3/2/pad1(relu(scale * conv7x7/2/pad3(x f32[B,3,H,W]) + shift)). */
long long di2p_stem_x3_packed_bytes(void);
int di2p_stem_x3_pack(const float* weight, void* Wp, void* stream);
int di2p_stem_x3_supported(int H, int W);
int di2p_stem_x3(const float* x, const void* Wp, const float* scale, const float* shift, float* y, int B, int H, int W, void* stream);
int di2p_global_avgpool(const float* x, float* y, int B, int C, int HW, void* stream);
/* out[b,c] = max_n x[b,c,n]  (networks_pc.py:115) */
int di2p_channel_max(const float* x, float* y, int B, int C, int N, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Registration: replaces evaluation/frustum_reg (registration.cpp:9-186 solvePGivenK, bound at
 * :190-206) + Ceres, and the 60-restart process fan-out of evaluation/registration_lsq.py:142-186.
 *
 * di2p_initial_guess: registration_lsq.py:196-220 on device.  points f64[F,3,N], labels
 *   i32[F,N] -> yaw0 f64[F]; labels_out i32[F,N] = label, or -1 for points removed by the front
 *   filter (the solver skips labels not in {0,1}, registration.cpp:87-125); has_inside i32[F].
 * di2p_solve_batched: F frames x R hypotheses.  One 4-wavefront workgroup per hypothesis; Cauchy-robust
 *   Levenberg-Marquardt with box bounds on t (see DESIGN.md for the exact algorithm statement).
 *   points f64[F,3,N], labels i32[F,N], K f64[F,3,3] (fx,fy,cx,cy used), init_y f64[F,R],
 *   init_T f64[F,R,3], lb/ub f64[3] HOST arrays, is_2d: 4 params [ry,tx,ty,tz] else 6
 *   [angle-axis, t].  Outputs: params f64[F,R,np], cost f64[F,R], iters i32[F,R], sweeps i32[F,R] (optional).
 *   If yaw0 != NULL it is added to init_y per frame (restart noise drawn before yaw0 is known).
 *   workspace: di2p_solve_workspace_bytes(F, R, N) bytes of scratch (sorted point records, cluster boxes, sort keys,
 *   parked Levenberg-Marquardt states of the two-tier launch).
 * di2p_select_best: argmin cost over R per frame (ties -> lowest r; frames with has_inside==0 get
 *   identity and cost 1e4, registration_lsq.py:329-332) -> best i32[F], P f64[F,4,4], cost f64[F].
 * di2p_solver_residuals: Problem::Evaluate of registration.cpp:150-155 at given params:
 *   loss-corrected residuals in point order (3 per label-1 point, 1 per label-0 point), compacted
 *   per frame into residuals f64[F, 3N] with counts i32[F]; cost f64[F]. */
int di2p_initial_guess(const double* points, const int32_t* labels, double* yaw0, int32_t* labels_out,
                       int32_t* has_inside, int F, int N, void* stream);
int di2p_solve_batched(const double* points, const int32_t* labels, const double* K,
                       const double* init_y, const double* init_T, const double* yaw0,
                       double H, double W, const double* lb_host, const double* ub_host,
                       int max_iter, int is_2d, int F, int R, int N,
                       double* params, double* cost, int32_t* iters, int32_t* sweeps /* may be NULL: #passes over the points */, void* workspace, void* stream);
/* Same solver reading the points as f32 [F,3,N] (the network's own pc tensor; widening to f64 is
 * exact, so results are bit-identical to di2p_solve_batched on the widened copy). */
int di2p_solve_batched_f32(const float* points, const int32_t* labels, const double* K,
                           const double* init_y, const double* init_T, const double* yaw0,
                           double H, double W, const double* lb_host, const double* ub_host,
                           int max_iter, int is_2d, int F, int R, int N,
                           double* params, double* cost, int32_t* iters, int32_t* sweeps /* may be NULL: #passes over the points */, void* workspace, void* stream);
long long di2p_solve_workspace_bytes(int F, int R, int N);
/* diagnostics only: device buffer of F*R*28 int64 (or NULL) receiving per-hypothesis phase cycle counts and cluster statistics
 * (layout: csrc/solver.hip at the definition; F*R*8 words up to version 2, 16 in version 3, 20 in versions 4-5).  While set, solves run
 * a separate, instrumented kernel. */
void di2p_solver_set_profile_buffer(void* buf);
int di2p_select_best(const double* params, const double* cost, const int32_t* has_inside, int is_2d,
                     int F, int R, int32_t* best, double* P, double* best_cost, void* stream);
int di2p_solver_residuals(const double* points, const int32_t* labels, const double* K,
                          const double* params, double H, double W, int is_2d, int F, int N,
                          double* residuals, int32_t* counts, double* cost, void* stream);

/* ---------------------------------------------------------------------------------------------
 * "Next" rows: the steps directly before and after the classifier.
 * di2p_farthest_point_sampling: FarthestSampler.sample of data/kitti_helper.py:224-243 (called at
 *   data/kitti_pc_img_pose_loader.py:416-423).  pts f32[B,3,M] (M <= 4096 candidate points), init_idx i32[B] or
 *   NULL (= 0; the reference draws randint(len(pts)) = randint(3)) -> idx i32[B,k], nodes f32[B,3,k] (may be NULL).
 *   numpy semantics: fp64 squared distances, np.minimum update, first-occurrence argmax.
 * di2p_gather_points: out[b,c,n] = src[b,c,idx[b,n]] -- the down-sampling of kitti_pc_img_pose_loader.py:158-171
 *   with the host-drawn index list as an explicit input.
 * di2p_project_labels: evaluation/visualize_and_save_data.py:100-115,138-139 (same arithmetic as
 *   models/multimodal_classifier.py:135-156): P f32[B,p_rows(3|4),4], K f32[B,3,3] -> coarse i32[B,N] (inside-frustum,
 *   <= W-1, z > 0.1), fine i32[B,N] = floor(px/scale) + floor(py/scale)*W_fine (may be NULL), pxpy f32[B,2,N] (may be NULL).
 * di2p_label_accuracy: :142-145 -> out f32[B,2] = {coarse accuracy, fine accuracy over gt-inside points (NaN if none)}.
 * di2p_pack_pc_label: the 7 x N hand-off record of :174-181 (xyz, coarse_pred, coarse_gt, fine_pred, fine_gt) as
 *   f64[B,7,N], i.e. what np.load(..._pc_label.npy) gives evaluation/registration_lsq.py:291-296. */
int di2p_farthest_point_sampling(const float* pts, const int32_t* init_idx, int32_t* idx_out, float* nodes_out,
                                 int B, int M, int k, void* stream);
int di2p_gather_points(const float* src, const int32_t* idx, float* out, int B, int C, int Nsrc, int Nout, void* stream);
int di2p_project_labels(const float* pc, const float* P, int p_rows, const float* K, float H, float W, float fine_scale,
                        int32_t* coarse, int32_t* fine, float* pxpy, int B, int N, void* stream);
int di2p_label_accuracy(const int32_t* coarse_pred, const int32_t* coarse_gt, const int32_t* fine_pred,
                        const int32_t* fine_gt, float* out, int B, int N, void* stream);
int di2p_pack_pc_label(const float* pc, const int32_t* coarse_pred, const int32_t* coarse_gt, const int32_t* fine_pred,
                       const int32_t* fine_gt, double* out, int B, int N, void* stream);

/* ---------------------------------------------------------------------------------------------
 * PnP back end of config 3 -- replaces cv2.solvePnPRansac as called by evaluation/registration_pnp.py:95-148
 * (solve_PnP).  OpenCV is absent (parity unpinned): the algorithm is stated in csrc/pnp.hip and DESIGN.md.
 *   pc f32[F,3,N], coarse i32[F,N] (1 = use the point), fine i32[F,N] (cell id: pixel = (fine % W_fine, fine / W_fine))
 *   or explicit pixels f32[F,2,N] (then fine may be NULL), K_scaled f64[F,3,3] (already multiplied by the 1/32 scale,
 *   camera_matrix_scaling :58-61), samples i32[F,iters,6] (RANSAC draws, taken modulo the frame's correspondence count),
 *   reproj_err in scaled pixels (reference: 0.6); local optimisation of the best model: refine_rounds rounds of
 *   {re-estimate inliers, refine_iters Gauss-Newton steps}, a round kept only if it loses no inliers.
 *   -> P f64[F,4,4] (identity when rejected), outlier_ratio f64[F] (1 when rejected), n_inliers, n_corr, best i32[F].
 *   workspace: di2p_pnp_workspace_bytes(F, N, iters). */
long long di2p_pnp_workspace_bytes(int F, int N, int iters);
/* The correspondence list alone: what solve_PnP (evaluation/registration_pnp.py:97-110) hands to cv2.solvePnPRansac (:125-127) -- points =
 * pc[:, coarse == 1], pixels = (fine - floor(fine / W) * W, floor(fine / W)), in point order.  corr f32[F][N][8] = {x, y, z, u, v, 0, 0, 0},
 * the first n_corr[f] records of frame f valid.  (The RANSAC entry points below pack internally; this one exists so the packing can be
 * compared with what the reference's own function passes to OpenCV: tests/golden/pnp_frontend_golden.npz.) */
int di2p_pnp_pack(const float* pc, const int32_t* coarse, const int32_t* fine, const float* pixels, int W_fine, int F, int N,
                  float* corr, int32_t* n_corr, void* stream);
int di2p_pnp_ransac(const float* pc, const int32_t* coarse, const int32_t* fine, const float* pixels,
                    const double* K_scaled, int W_fine, const int32_t* samples, int iters, double reproj_err,
                    int refine_rounds, int refine_iters, int F, int N, double* P_out, double* outlier_ratio, int32_t* n_inliers,
                    int32_t* n_corr, int32_t* best, void* workspace, void* stream);
/* The estimator the reference names (flags = cv2.SOLVEPNP_EPNP, evaluation/registration_pnp.py:125-132): EPnP on minimal
 * samples of 5 correspondences (the first 5 entries of each row of `samples`; 4 when the frame has only 4), inlier iff the squared
 * reprojection error <= reproj_err^2, the model with the most inliers, one EPnP re-fit on all of its inliers; frames with
 * fewer than 4 correspondences or no model are rejected (identity, outlier ratio 1).  Same buffers as di2p_pnp_ransac. */
int di2p_pnp_ransac_epnp(const float* pc, const int32_t* coarse, const int32_t* fine, const float* pixels,
                         const double* K_scaled, int W_fine, const int32_t* samples, int iters, double reproj_err, int F, int N,
                         double* P_out, double* outlier_ratio, int32_t* n_inliers, int32_t* n_corr, int32_t* best,
                         void* workspace, void* stream);

/* f32 -> f64 widening copy of the point cloud for the solver ([B,3,N]) and i32 label passthrough
 * are done by the caller; helper for the fused pipeline: */
int di2p_f32_to_f64(const float* in, double* out, long long n, void* stream);

/* ---- on-device random draws (counter-based Philox4x32-10: every value is a function of (seed, index) only) -------------
 * di2p_draw_restarts: the solver's restart list (evaluation/registration_lsq.py:163-164): ry_noise f64[F,R] ~ N(0, ry_sigma)
 *   (the frame's yaw0 is added by the solver), init_T f64[F,R,3] = (0, 0, U(-t_amplitude, t_amplitude)).
 * di2p_random_choice: n_out of n_src indices without replacement, in random order, per frame
 *   (np.random.choice(n_src, n_out, replace=False) of data/kitti_pc_img_pose_loader.py:158-171,416-423);
 *   stream_id separates independent draws under one seed; workspace: di2p_random_choice_workspace_bytes(B, n_src). */
int di2p_draw_restarts(unsigned long long seed, int F, int R, double ry_sigma, double t_amplitude, double* ry_noise,
                       double* init_T, void* stream);
long long di2p_random_choice_workspace_bytes(int B, int n_src);
int di2p_random_choice(unsigned long long seed, int stream_id, int B, int n_src, int n_out, int32_t* idx_out, void* workspace,
                       void* stream);

/* ---- training-side head (losses and optimiser; the backward kernels follow below) ----------------
 * di2p_classifier_loss: the losses of models/multimodal_classifier.py:189-191 and d loss / d scores in one pass:
 *   coarse f32[B,2,N] with FocalLoss(alpha, gamma, 'mean') * coarse_loss_alpha (models/focal_loss.py:55-112), fine f32[B,L,N]
 *   (or NULL) with mean cross-entropy over the points whose coarse label is 1; labels i32[B,N] (di2p_project_labels).
 *   out8 f64[8] = {loss, coarse loss, fine loss, coarse accuracy, fine accuracy, inside count, 0, 0};
 *   d_coarse f32[B,2,N] / d_fine f32[B,L,N] (may be NULL: losses only).  workspace: di2p_classifier_loss_workspace_bytes(B, N).
 * di2p_adam_step: torch.optim.Adam (weight_decay 0, amsgrad off; :44-47) on one flat fp32 buffer, step counted from 1. */
long long di2p_classifier_loss_workspace_bytes(int B, int N);
int di2p_classifier_loss(const float* coarse, const float* fine, const int32_t* coarse_labels, const int32_t* fine_labels,
                         int B, int N, int L, float alpha, float gamma, float coarse_loss_alpha, double* out8,
                         float* d_coarse, float* d_fine, void* workspace, void* stream);
int di2p_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int step, float lr,
                   float beta1, float beta2, float eps, void* stream);

/* ---- training: backward kernels and train-mode BatchNorm (replaces the torch.autograd backward of
 *      models/multimodal_classifier.py:213-218).  All tensors f32, [B,C,N] (or NCHW with N = H*W), contiguous. ----------------------
 * di2p_bn_train_forward: nn.BatchNorm1d/2d in train mode (layers_pc.py:283,150-160; resnet.py:44-70): batch mean / biased variance over
 *   (B,N), y = relu?((x-mean)*invstd*gamma + beta + residual?); running stats updated with `momentum` and the unbiased variance
 *   (running_* may both be NULL).  workspace: di2p_channel_reduce_workspace_bytes(B, C, N).
 * di2p_bn_train_backward: g = dy * [y > 0] (relu) ; dgamma = sum g*xhat, dbeta = sum g, dx = gamma*invstd*(g - mean g - xhat*mean(g*xhat)),
 *   dresidual = g (may be NULL).
 * di2p_channel_sum: out[c] = sum_{b,n} x[b,c,n] (bias gradients).
 * di2p_bmm_rc: out[z][row][col] = alpha * sum_r A[z][row*lda + r] * B[z][col*ldb + r]  (both operands contiguous along the reduction:
 *   d W[M][K] = sum_{b,n} dY[b][m][n] X[b][k][n] with reduce_z = 1; attention d feat with reduce_z = 0).  Deterministic chunked reduction;
 *   workspace: di2p_bmm_rc_workspace_bytes(Z, rows, cols, R).
 * di2p_bmm_km: out[z][row][col] = alpha * sum_k A[z][k*lda + row] * B[z][k*ldb + col]  (attention d score).
 * di2p_gather_backward: d feats[b,c,m] = sum_j dy[b,c,j] * sum_k w[b,j,k] * [idx[b,j,k] == m]; k = 1 (torch.gather along the point axis,
 *   weights NULL = 1) or k = 3 (upsample_by_interpolation, networks_united.py:90-103).  workspace: di2p_gather_backward_workspace_bytes.
 * di2p_conv2d_wgrad / di2p_conv2d_dgrad: gradients of nn.Conv2d (bias-free; resnet.py) w.r.t. weight[Cout][Cin][KH][KW] and input.
 * di2p_maxpool3x3s2_backward, di2p_segment_max_backward (index_max + gather + mask, networks_pc.py:88-93), di2p_group_max_forward /
 *   _backward (max over the last axis with its first arg-max): the gradient goes to the arg-max element.
 * di2p_dropout_mask / di2p_apply_mask: nn.Dropout keep-mask from the counter-based generator (a function of seed, stream_id, index)
 *   and y = mask ? x*scale : 0. */
long long di2p_channel_reduce_workspace_bytes(int B, int C, int N);
int di2p_bn_train_forward(const float* x, const float* gamma, const float* beta, const float* residual, float* y, float* save_mean,
                          float* save_invstd, float* running_mean, float* running_var, float momentum, float eps, int relu, int B, int C,
                          int N, void* workspace, void* stream);
int di2p_bn_train_backward(const float* x, const float* y, const float* dy, const float* gamma, const float* save_mean,
                           const float* save_invstd, int relu, float* dx, float* dresidual, float* dgamma, float* dbeta, int B, int C, int N,
                           void* workspace, void* stream);
int di2p_channel_sum(const float* x, float* out, int B, int C, int N, void* workspace, void* stream);
long long di2p_bmm_rc_workspace_bytes(int Z, int rows, int cols, int R);
int di2p_bmm_rc(const float* A, long long lda, long long a_batch_stride, const float* Bm, long long ldb, long long b_batch_stride, float* out,
                int Z, int rows, int cols, int R, float alpha, int reduce_z, void* workspace, long long workspace_bytes, void* stream);
int di2p_bmm_km(const float* A, int lda, long long a_batch_stride, const float* Bm, int ldb, long long b_batch_stride, float* out, int Z,
                int rows, int cols, int K, float alpha, void* stream);
long long di2p_gather_backward_workspace_bytes(int B, int C, int J, int M);
int di2p_gather_backward(const float* dy, const int32_t* idx, const float* weights, int k, float* dfeats, int B, int C, int J, int M,
                         void* workspace, long long workspace_bytes, void* stream);
long long di2p_conv2d_wgrad_workspace_bytes(int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad);
int di2p_conv2d_wgrad(const float* x, const float* dy, float* dW, int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                      void* workspace, long long workspace_bytes, void* stream);
int di2p_conv2d_dgrad(const float* dy, const float* Wgt, float* dx, int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                      void* stream);
int di2p_maxpool3x3s2_backward(const float* x, const float* dy, float* dx, int B, int C, int H, int W, void* stream);
int di2p_segment_max_backward(const float* dvalues, const int32_t* max_idx, const float* mask, float* dx, int B, int C, int N, int M,
                              void* stream);
int di2p_group_max_forward(const float* x, float* y, int32_t* arg, long long rows, int K, void* stream);
int di2p_group_max_backward(const float* dy, const int32_t* arg, float* dx, long long rows, int K, void* stream);
int di2p_dropout_mask(unsigned long long seed, int stream_id, float p, long long n, uint8_t* mask, void* stream);
int di2p_apply_mask(const float* x, const uint8_t* mask, float scale, float* y, long long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPI2P_HIP_H */
