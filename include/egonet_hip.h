/*
 * egonet_hip.h -- C ABI of the MI355X (gfx950) hot-path library of egonet_amd.
 *
 * The reference (Nicholasli1995/EgoNet) is pure Python on torch.nn: it has no
 * FFI of its own.  Each entry point below replaces the ATen op sequence issued
 * by the cited reference lines; the Python host layer (egonet_amd/model/...)
 * mirrors the reference's operator API (get_pose_net / get_fc_model / EgoNet)
 * and reaches these functions through ctypes.
 *
 * Conventions
 *   - every function returns int: 0 = ok, >0 = hipError_t, <0 = EGN_E_* below;
 *     no exception crosses this boundary.
 *   - all pointers are DEVICE pointers owned by the caller (torch tensors);
 *     the library never allocates user-visible memory.
 *   - `stream` is a hipStream_t passed as void*; nothing synchronises.
 *   - activations are fp32 NHWC with a channel stride `cs` (multiple of 4,
 *     pad channels hold zeros) unless a parameter says NCHW.
 */
#ifndef EGONET_HIP_H
#define EGONET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGN_E_BADARG   (-1)   /* inconsistent shapes / unsupported parameter  */
#define EGN_E_LDS      (-2)   /* tile does not fit in the 160 KiB LDS          */
#define EGN_E_STATE    (-3)   /* program used in the wrong state               */

#define EGN_ACT_NONE    0
#define EGN_ACT_RELU    1
#define EGN_ACT_SIGMOID 2
#define EGN_ACT_LEAKY   3     /* slope 0.01 (torch LeakyReLU default)          */
#define EGN_ACT_MASK    0x0f
#define EGN_ACT_RES_AFTER 0x10 /* flag: y = res + act(..) instead of act(.. + res)
                                  (FCmodel.py:34-43: out = x + relu(bn(w2(..))))  */

/* library / ABI version: (major<<16)|minor */
int egn_version(void);
/* human readable text for a return code (static storage) */
const char* egn_strerror(int code);

/* ------------------------------------------------------------------------
 * Convolution as fp32-MFMA implicit GEMM, fused  y = act(conv(x)*scale+shift
 * (+res)).  Replaces Conv2d+BatchNorm2d(+add)+ReLU chains:
 *   hrnet.py:76-92 (BasicBlock), :113-133 (Bottleneck), :232-274 (fuse convs),
 *   :482-507 (transitions), :318-323 (stem), :365-371/:427-435 (1x1 heads),
 *   :457 (4x4 valid conv + Sigmoid); FCmodel.py:33-43,92-105 (Linear+BN1d as
 *   1x1 conv on [N,1,1,C]).
 *
 * x      [N,H,W,cs_in]  fp32 NHWC
 * wpack  weights prepacked by egn_conv_pack_size/egonet_amd.engine.pack_conv:
 *        [nchunk][KH*KW][CK/4][CoutP][4] fp32, CK = 16 input channels per chunk,
 *        CoutP = Cout rounded up to 16, zero padded
 * scale, shift  [CoutP] fp32 (folded BatchNorm / bias), zero padded
 * res    optional residual [N,Ho,Wo,cs_out] (NULL = none), added before act
 * y      [N,Ho,Wo,cs_out] NHWC, or [N,Cout,Ho,Wo] when out_nchw != 0
 * cfg    tile configuration id (0 = choose automatically)
 * ---------------------------------------------------------------------- */
int egn_conv2d_f32(const float* x, const float* wpack, const float* scale,
                   const float* shift, const float* res, float* y,
                   int N, int H, int W, int Cin, int cs_in,
                   int Cout, int cs_out, int KH, int KW, int stride, int pad,
                   int act, int out_nchw, int cfg, void* stream);

/* Host-only: the launch plan egn_conv2d_f32 would use (no GPU needed).
 * out[0..11] = cfg, wm, wn, mt, nt, TH, TW, TNB, taps_per_stage, lds_bytes,
 *              grid_x, grid_y */
int egn_conv_plan_query(int N, int H, int W, int Cin, int cs_in, int Cout,
                        int cs_out, int KH, int KW, int stride, int pad,
                        int out_nchw, int cfg, int* out);

/* number of tile configurations compiled in; valid ids are 1..count */
int egn_conv_num_configs(void);
/* describe config id: tile_m (output pixels), tile_n (output channels) */
int egn_conv_config_info(int cfg, int* tile_m, int* tile_n);
/* kernel symbol of config id as rocprofv3 prints it, NUL terminated into buf */
int egn_conv_config_name(int cfg, char* buf, int len);
/* which filter packing config id expects in `wpack`:
 *   0  direct kernels: egn_pack_conv_weight_f32 (layout above)
 *   1  fused Winograd F(2x2,3x3) kernels (3x3, stride 1, pad 1, Cin % 16 == 0,
 *      Cout % 48 == 0 or -- the 8-wave kernels -- Cout % 32 == 0, unpadded channel
 *      strides, act none/ReLU): the TRANSFORMED filter from egn_wino_pack_weight_f32
 *   2  fused Winograd F(4x4,3x3), conv_wino43_kernel: U = G g G^T for points 0, +-1, +-2,
 *      inf packed [Cout/48][Cin/4][f = 6i+j][ci % 4][48] (host: engine.pack_wino43_weight)
 *   3  fused Winograd F(4x4,3x3), conv_wino4_kernel / conv_wino4b_kernel / conv_wino4c_kernel (3x3,
 *      stride 1, pad 1, Cin % 16 == 0, Cout % 48 == 0, maps of whole 16 x 32 / 16 x 16 pixel regions
 *      or 8 x 8 maps): the same U packed for register feeding, [Cout/48][k-group Cin/4][wave 12][3]
 *      [64 lanes][4] floats (9 of 12 used) (egn_wino4_weight_floats; host: engine.pack_wino4_weight).
 *      Configs 83 / 84 (8 x 8 maps / 16 x 16 regions, input channels of a work item split over two
 *      blocks; Cin % 32 == 0):
 *      called through egn_conv2d_f32 it is THREE launches on `stream` (y zeroed, atomic adds, in-place
 *      epilogue; res must not alias y); as an op of a program it is one launch.
 *  -1  not selectable: invalid id, retired family, timing-ablation / stamp build.  The product
 *      library neither plans nor launches such an id (egn_conv2d_f32 returns EGN_E_BADARG);
 *      a probe build (-DEGN_PROBES, tools/ only) compiles them. */
int egn_conv_config_kind(int cfg);
/* 1 = this library is a probe build (python -m egonet_amd.build --probes), 0 = the product */
int egn_probe_build(void);
/* floats of the kind-3 filter of a [Cout][Cin][3][3] weight (0 = shape not supported) */
long long egn_wino4_weight_floats(int Cout, int Cin);
/* The kind-3 filter transform on the device: torch weight [Cout][Cin][3][3] -> U = G g G^T (float64
 * arithmetic, rounded once to fp32) in the register-feed layout of kind 3 -- the device form of
 * engine.pack_wino4_weight; dgrad 1 = the data-gradient filter (channels swapped, taps rotated by
 * 180 degrees).  egn_wino4_pack_weight_floats = floats dst must hold (0 = shape not supported:
 * output channels % 48, input channels % 8, >= 16).  Replaces nothing in the reference (cuDNN
 * transforms filters internally for the Conv2d calls of hrnet.py:24-27); it is what lets weights
 * that change every step (libs/trainer/trainer.py:183-209) feed the F(4x4,3x3) kernels. */
long long egn_wino4_pack_weight_floats(int Cout, int Cin, int dgrad);
int egn_wino4_pack_weight_f32(const float* w, int Cout, int Cin, int dgrad, float* dst, void* stream);
/* Winograd filter transform on the device: torch weight [Cout][Cin][3][3] ->
 * U = G g G^T (float64 arithmetic, rounded once to fp32) packed as
 * [Cout/T][Cin/16][f = 4i+j][quad][T][4] floats (ci = chunk*16 + quad*4 + r),
 * T = 48 if Cout % 48 == 0 else 32 (Cout % 32 == 0): one contiguous slab per
 * (output-channel tile, input-channel chunk).
 * dgrad 1 = the data-gradient filter (channels swapped, taps rotated 180 deg).
 * egn_wino_weight_floats = floats dst must hold (0 = shape not supported).
 * Replaces nothing in the reference (cuDNN picks its Winograd algorithms
 * internally for the Conv2d calls of hrnet.py:24-27); it is the packing step of
 * this library's 3x3 kernels. */
long egn_wino_weight_floats(int Cout, int Cin, int dgrad);
int egn_wino_pack_weight_f32(const float* w, int Cout, int Cin, int dgrad,
                             float* dst, void* stream);

/* ------------------------------------------------------------------------
 * Multi-resolution fuse: y = relu( ((t0 + up(t1)) + up(t2)) + up(t3) ),
 * left-associated in the order given (hrnet.py:291-298); up() = nearest
 * neighbour by 2^shift (hrnet.py:241), shift 0 = same resolution.
 * All tensors NHWC with channel stride cs; y is [N,H,W,cs]; t_i is
 * [N,H>>s_i,W>>s_i,cs].  nterms in 1..4.
 * ---------------------------------------------------------------------- */
int egn_fuse_sum_relu_f32(float* y, int N, int H, int W, int C, int cs,
                          int nterms, const float* const* terms,
                          const int* shifts, int relu, void* stream);

/* layout helpers at the module boundary (reference tensors are NCHW) */
int egn_nchw_to_nhwc_f32(const float* x, float* y, int N, int C, int H, int W,
                         int cs, void* stream);
int egn_nhwc_to_nchw_f32(const float* x, float* y, int N, int C, int H, int W,
                         int cs, void* stream);
/* nn.PixelShuffle(up) of the heat-map upsampler (hrnet.py:373-383, 598-600) fused with
 * the NHWC -> NCHW hand-over: y[n,j,h*up+a,w*up+b] = x[n,h,w,(j*up+a)*up+b];
 * x [N,H,W,cs] with cs >= C*up*up, y [N,C,H*up,W*up] */
int egn_pixel_shuffle_nhwc_to_nchw_f32(const float* x, float* y, int N, int C,
                                       int H, int W, int cs, int up, void* stream);
/* write the two coordinate ramps linspace(0,1) (hrnet.py:461-467) into
 * channels [c0, c0+1] of an NHWC tensor */
int egn_fill_coord_ramps_f32(float* y, int N, int H, int W, int cs, int c0,
                             void* stream);

/* ------------------------------------------------------------------------
 * Key-point decode, one wavefront per (n,k) map (img_proc.py:608-637 hard
 * arg-max, :678-707 soft-arg-max).  hm is NCHW [N,K,H,W].
 *   out_xy   [N,K,2] fp32   (x,y) in heat-map pixels
 *   out_max  [N,K]   fp32   raw maximum
 *   out_idx  [N,K]   int32  flat arg-max index (first max on ties); may be NULL
 * mode 0 = hard arg-max (coordinates zeroed where max <= 0),
 * mode 1 = soft-arg-max (softmax over H*W, no mask),
 * mode 2 = soft_arg_max_np (img_proc.py:639-676: weights hm/sum(hm),
 *          coordinates zeroed where max <= 0).
 * ---------------------------------------------------------------------- */
int egn_decode_heatmaps_f32(const float* hm, int N, int K, int H, int W,
                            int mode, float* out_xy, float* out_max,
                            int32_t* out_idx, void* stream);

/* ------------------------------------------------------------------------
 * Crop-local key-points -> screen coordinates -> normalised lifter input
 * (egonet.py:436-453 + img_proc.py:26-78 with rot=0, operations.py:20-47).
 *   local   [n,K,2] fp32, multiplied by (mul_x, mul_y) first
 *           (coords head: resolution; soft-arg-max: input/heatmap size)
 *   center  [n,2] f64, scale [n,2] f64 (modify_bbox outputs), crop size (cw,ch)
 *   screen  [n,K*2] f64 out (records['kpts_2d_pred'])
 *   mean_in, std_in [K*2] f64;  lifter_in [n, ld_in] fp32 out (may be NULL)
 * ---------------------------------------------------------------------- */
int egn_keypoints_to_screen_f64(const float* local, int n, int K,
                                double mul_x, double mul_y,
                                const double* center, const double* scale,
                                int crop_w, int crop_h, double* screen,
                                const double* mean_in, const double* std_in,
                                float* lifter_in, int ld_in, void* stream);

/* pred3d[n,D] f64 = y[n,ld] fp32 * std_out + mean_out  (operations.py:49-51) */
int egn_unnormalize_f64(const float* y, int n, int D, int ld,
                        const double* mean_out, const double* std_out,
                        double* pred3d, void* stream);

/* ------------------------------------------------------------------------
 * Batched pose solve (egonet.py:238-295, transformation.py:99-134,
 * egonet.py:203-236): cuboid template from mean edge lengths, Kabsch
 * rotation by 3x3 SVD, euler 'yxz' -> (x,y,z) order, observation angle.
 *   pred3d [n,32,3] f64;  kpt_x [n] f64 = screen x of key-point 0 (proj mode)
 *   euler [n,3] f64, alpha [n] f64;  alpha_mode 0 = 'proj' (needs fx,cx),
 *   1 = 'trans'
 * ---------------------------------------------------------------------- */
int egn_pose_solve_f64(const double* pred3d, int n, const double* kpt_x,
                       double fx, double cx, int alpha_mode,
                       double* euler, double* alpha, void* stream);
/* Host twins of egn_keypoints_to_screen_f64 (without the lifter-input part) and
 * egn_pose_solve_f64: the same arithmetic as plain loops over HOST pointers, no
 * stream.  They serve the reference's CPU plumbing -- EgoNet.get_keypoints(
 * is_cuda=False), get_6d_rep on a CPU model (egonet.py:424-467, 279-295;
 * BASELINE config 1) -- and are never used for CUDA tensors. */
int egn_keypoints_to_screen_host_f64(const float* local, int n, int K,
                                     double mul_x, double mul_y,
                                     const double* center, const double* scale,
                                     int crop_w, int crop_h, double* screen);
int egn_pose_solve_host_f64(const double* pred3d, int n, const double* kpt_x,
                            double fx, double cx, int alpha_mode,
                            double* euler, double* alpha);

/* ------------------------------------------------------------------------
 * KITTI 2D-detection AP + Average Orientation Similarity, the IMAGE-metric path
 * of tools/kitti-eval/evaluate_object_3d_offline.cpp (:131-265, :346-706,
 * :791-850) without Boost.  Host code, no GPU.  Reads gt_dir/%06d.txt and
 * result_dir/data/%06d.txt (KITTI label / result lines).
 *   evaluated [3]       1 if class (car, pedestrian, cyclist) has detections
 *   aos_valid           0 if any detection carries alpha == -10
 *   precision, aos      [3 classes][3 levels easy/moderate/hard][41 recall samples]
 * Returns 0; -1 bad argument; -2 missing ground-truth file; -3 no result files.
 * ---------------------------------------------------------------------- */
int egn_kitti_eval_image(const char* gt_dir, const char* result_dir, int* n_frames,
                         int* evaluated, int* aos_valid, double* precision,
                         double* aos);

/* ------------------------------------------------------------------------
 * Crop front end: all boxes of one image in one launch (egonet.py:68-96:
 * get_affine_transform + cv2.warpAffine(INTER_LINEAR, border 0) + ToTensor +
 * Normalize).  img [H,W,3] uint8 RGB (row pitch in bytes); M [n,6] f64 forward
 * 2x3 affines image -> crop; out [n,3,out_h,out_w] f32 = (bilinear/255 - mean)/std.
 * Restates cv::warpAffine's 8-bit fixed-point scheme (1/32 pixel); OpenCV is
 * absent from the build image: parity with cv2 is unpinned.
 * ---------------------------------------------------------------------- */
int egn_crop_warp_normalize_u8(const uint8_t* img, int H, int W, int pitch,
                               const double* M, int n, int out_h, int out_w,
                               const float* mean, const float* stdv, float* out,
                               void* stream);

/* ------------------------------------------------------------------------
 * Training step building blocks (reference: libs/trainer/trainer.py:183-209
 * zero_grad / forward / loss / backward / optim.step; FCmodel.py Linear +
 * BatchNorm1d (batch statistics) + ReLU + Dropout; function.py:204-215
 * MSELoss1D; optimizer.py:8-40 Adam).  The GEMMs of forward / dgrad / wgrad run
 * on egn_conv2d_f32 (a Linear is a 1x1 conv); matrices are row-major
 * [rows, ld] fp32 with ld % 4 == 0.
 * ---------------------------------------------------------------------- */
/* Weight gradient of a conv / Linear layer (autograd's backward of nn.Conv2d /
 * nn.Linear in trainer.py:194 `loss.backward()`):
 *   dw[co][ci][ky][kx] = sum_{n,oy,ox} dy[n,oy,ox,co] * x[n,oy*s+ky-p,ox*s+kx-p,ci]
 * x [N,H,W,cs_in], dy [N,Ho,Wo,cs_out] NHWC fp32 (pad channels zero), dw in
 * torch's [Cout][Cin][KH][KW] layout.  Split-K over pixel tiles with a
 * deterministic two-pass reduction; ws = scratch of egn_conv2d_wgrad_ws_bytes().
 * KH*KW in {1, 9, 16}.  A Linear is N = rows, H = W = 1.
 * 3x3 / stride 1 / pad 1 layers with Cin % 48 == 0, Cout % 48 == 0 and unpadded
 * channel strides run in Winograd F(2x2,3x3) form (csrc/conv_wgrad_wino.hip: same
 * result to fp32 rounding -- error vs float64 4e-7 of the largest tap, the direct
 * kernel's 3e-7 -- not bit-identical; the environment switch EGN_WGRAD_WINO=0
 * keeps the direct kernel).  Either way two launches give the same bits. */
long egn_conv2d_wgrad_ws_bytes(int N, int H, int W, int Cin, int cs_in, int Cout,
                               int cs_out, int KH, int KW, int stride, int pad);
int egn_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, int N, int H,
                         int W, int Cin, int cs_in, int Cout, int cs_out, int KH,
                         int KW, int stride, int pad, void* ws, long ws_bytes,
                         void* stream);
/* conv weights [nchunk][1][4][CoutP][4] from a row-major matrix:
 * transpose 0: W[co][ci] = src[co*ld+ci]; 1: W[co][ci] = src[ci*ld+co] */
int egn_pack_matrix_f32(const float* src, int ld, int cout, int cin,
                        int transpose, float* dst, void* stream);
/* Dense fp32 GEMM on the matrix pipe (csrc/gemm.hip) for the lifter's nn.Linear layers -- reference
 * libs/model/FCmodel.py:33-43, 92-105 (forward) and what torch autograd derives from them in
 * libs/trainer/trainer.py:191-197 (data / weight gradients); operands are read as they lie, nothing is packed:
 *   form 0 "NT"  C[M][N] = A[M][K] . B[N][K]^T (+ bias[N])     z  = a W^T + b
 *   form 1 "NN"  C[M][N] = A[M][K] . B[K][N]                   da = dz W
 *   form 2 "TN"  C[M][N] = A[K][M]^T . B[K][N]                 dW = dz^T a   (split along K, fixed-order reduction;
 *                                                              `ws` of egn_gemm_ws_bytes bytes)
 * Row-major with leading dimensions lda / ldb / ldc (floats, multiples of 4).  egn_gemm_supported() = 1 for the
 * shapes the kernels take (M, N multiples of 128, K of 32); the callers keep the conv-kernel route for the rest.
 * variant 0 = default tile configuration of the form (others: tools/gemm_probe.py). */
int egn_gemm_supported(int form, int M, int N, int K, int lda, int ldb, int ldc);
long egn_gemm_ws_bytes(int form, int M, int N, int K);
int egn_gemm_f32(int form, const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                 int lda, int ldb, int ldc, int variant, void* ws, long ws_bytes, void* stream);
/* The same with the epilogue fusions of the lifter's training step (FCmodel.py:33-43 under trainer.py:183-209):
 *   addend (form 1 only, [M][ldc]): C = A.B + addend -- the skip-path gradient of a residual block (FCmodel.py:49-51,
 *           `out = x + y`) joins the branch gradient without an add pass;
 *   stats  (form 0 only): partial column sums and sums of squares of the stored C, [stats_rows][2][N]
 *           doubles with stats_rows >= egn_gemm_stats_rows(M) (one row per 128-row block tile, fixed summation order)
 *           -- nn.BatchNorm1d's batch statistics without a pass over z; egn_bn_stats_finalize_f32 adds the rows. */
long egn_gemm_stats_rows(int M);
int egn_gemm_ex_f32(int form, const float* A, const float* B, float* C, const float* bias, const float* addend,
                    double* stats, long stats_rows, int M, int N, int K, int lda, int ldb, int ldc, int variant,
                    void* ws, long ws_bytes, void* stream);
/* dst[c][r] = src[r][c]; dst columns R..ld_dst-1 are zeroed */
int egn_transpose_f32(const float* src, int R, int C, int ld_src, float* dst,
                      int ld_dst, void* stream);
/* scratch bytes the column reductions below need for `cols` columns */
long egn_colreduce_ws_bytes(int cols);
int egn_colsum_f32(const float* a, int rows, int cols, int ld, float* sum,
                   void* ws, void* stream);
/* per-column batch statistics of z [rows, ld] (NHWC conv output: rows = N*H*W,
 * ld = cs): mean, 1/sqrt(biased var + eps), unbiased var; when running_mean /
 * running_var are given they are updated in place with torch's rule
 * running = (1-momentum)*running + momentum*batch (unbiased var) */
int egn_bn_stats_f32(const float* z, int rows, int cols, int ld, float eps,
                     float* mean, float* invstd, float* var_unbiased,
                     float* running_mean, float* running_var, float momentum,
                     void* ws, void* stream);
/* BatchNorm batch statistics WITHOUT a second pass over the conv output: the raw convolution
 * (scale = ones, shift = zeros, no residual / activation; `wpack` as egn_conv_config_kind(cfg) says)
 * whose epilogue also writes, per (tile, wave), the column sums and sums of squares of what it stores:
 * partials [rows][2][Cout] doubles, rows = egn_conv2d_bnstats_rows(...) (0 = this tile configuration
 * has no fused statistics: use egn_conv2d_f32 + egn_bn_stats_f32).  egn_bn_stats_finalize_f32 adds
 * the rows in a fixed order (deterministic) and produces what egn_bn_stats_f32 produces
 * (`rows` there = N*Ho*Wo, the number of samples per channel). */
long egn_conv2d_bnstats_rows(int N, int H, int W, int Cin, int cs_in, int Cout,
                             int cs_out, int KH, int KW, int stride, int pad, int cfg);
int egn_conv2d_bnstats_f32(const float* x, const float* wpack, const float* ones,
                           const float* zeros, float* y, int N, int H, int W,
                           int Cin, int cs_in, int Cout, int cs_out, int KH, int KW,
                           int stride, int pad, int cfg, double* partials,
                           long partial_rows, void* stream);
/* [round 5] The 1x1 convolutions around layer1's 256-channel tensor as ONE streaming launch (csrc/conv_pw.hip).
 * Replaces, for a Bottleneck of libs/model/heatmapModel/hrnet.py:95-133 and the first line of the next one,
 *     out = relu(bn3(conv3(h)) + residual)    (1x1, 64 -> 256)    and    hn = relu(bn1(conv1(out)))    (1x1, 256 -> 64):
 *     out[M][256] = act1(h[M][64] . w3^T + shift3 (+ res[M][256])),    hn[M][64] = relu(out . w1^T + shift1)
 * with the block's tile of `out` kept in LDS between the two products (the 256-channel tensor crosses HBM once per
 * direction).  w3 [256][64] / w1 [64][256]: the 1x1 filters as torch holds them ([Cout][Cin], row-major) with the
 * folded BatchNorm scale multiplied in per output channel; shift: the folded shift.  act1 = ReLU if relu1 else none.
 * w1 = shift1 = hn = NULL: the first product alone (the downsample conv, the last block's conv3).
 * M = N*H*W pixels, M % 32 == 0; NHWC rows without padding; res != out, h != hn. */
int egn_pw_pair_f32(const float* h, const float* res, const float* w3, const float* shift3,
                    const float* w1, const float* shift1, float* out, float* hn, int M,
                    int relu1, void* stream);
/* [round 5] egn_conv2d_f32 with the two optional side tables of the native training tape
 * (replaces the MIOpen convolutions under libs/trainer/trainer.py:191-197 for the 3x3 s1 layers of
 * libs/model/heatmapModel/hrnet.py:63-92 on the F(4x4,3x3) kernels):
 *   partials / partial_rows  BatchNorm partial statistics of the stored output, as egn_conv2d_bnstats_f32
 *                            (NULL = none; else egn_conv2d_bnstats_rows(cfg) rows are written);
 *   tickets / ticket_words   zeroed `unsigned` words for the K-split configurations (cfg 83 / 84):
 *                            egn_conv2d_ticket_words(...) of them (0 = the configuration uses none).  With
 *                            them the layer is ONE kernel launch, as inside a program, and leaves the words
 *                            zero; launches that share the words must be ordered (one stream).  NULL = the
 *                            three-launch form of egn_conv2d_f32. */
long egn_conv2d_ticket_words(int N, int H, int W, int Cin, int cs_in, int Cout,
                             int cs_out, int KH, int KW, int stride, int pad, int cfg);
int egn_conv2d_ex_f32(const float* x, const float* wpack, const float* scale,
                      const float* shift, const float* res, float* y, int N, int H,
                      int W, int Cin, int cs_in, int Cout, int cs_out, int KH, int KW,
                      int stride, int pad, int act, int cfg, double* partials,
                      long partial_rows, unsigned* tickets, long ticket_words,
                      void* stream);
int egn_bn_stats_finalize_f32(const double* partials, long nrows, int rows, int cols,
                              float eps, float* mean, float* invstd,
                              float* var_unbiased, float* running_mean,
                              float* running_var, float momentum, void* stream);
/* relu: 0 none, 1 nn.ReLU, 2 nn.LeakyReLU() (slope 0.01, FCmodel.py:19-22) -- here and in the two
 * backward entry points below.
 * y = relu?(gamma*(z-mean)*invstd + beta + res?) * (mask ? mask*keep_scale : 1)
 * (BatchNorm on batch statistics + residual + ReLU + inverted dropout) */
int egn_bn_act_fwd_f32(const float* z, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, const float* mask,
                       float keep_scale, int relu, const float* res, float* y,
                       int rows, int cols, int ld, void* stream);
/* BatchNorm backward, step 1: dbeta = sum dpre, dgamma = sum dpre*xhat with
 * dpre = dy * dropout-mask * relu-gate(gamma*xhat + beta + res > 0) */
int egn_bn_bwd_sums_f32(const float* dy, const float* z, const float* mask,
                        float keep_scale, const float* mean, const float* invstd,
                        const float* gamma, const float* beta, int relu,
                        const float* res, int rows, int cols, int ld,
                        float* dbeta, float* dgamma, void* ws, void* stream);
/* step 2: dz = gamma*invstd*(dpre - dbeta/rows - xhat*dgamma/rows); dres
 * (optional) = dpre, the gradient that flows into the residual branch */
int egn_bn_bwd_dz_f32(const float* dy, const float* z, const float* mask,
                      float keep_scale, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, int relu,
                      const float* res, const float* dbeta, const float* dgamma,
                      float* dz, float* dres, int rows, int cols, int ld,
                      void* stream);
/* The same three with the inverted-dropout keep mask DRAWN IN THE KERNEL (nn.Dropout(p) after every ReLU of the
 * lifter, libs/model/FCmodel.py:24, 38-41): Philox4x32-10 keyed by `seed`, counter = (float4 group index of the
 * element, `layer`, *step_dev); element kept <=> its 32-bit draw >= p * 2^32; kept values are scaled by 1/(1-p).
 * Forward and backward regenerate the same mask from the same (seed, layer, *step_dev): no mask tensor, no RNG
 * kernel inside a training step; *step_dev is read on the device (the optimizer's step counter: hipGraph-safe).
 * egn_dropout_mask_f32 writes the mask those kernels use (tests). */
int egn_bn_act_fwd_drop_f32(const float* z, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, float p, unsigned long long seed, const int* step_dev, int layer,
                            int relu, const float* res, float* y, int rows, int cols, int ld, void* stream);
int egn_bn_bwd_sums_drop_f32(const float* dy, const float* z, float p, unsigned long long seed, const int* step_dev,
                             int layer, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, int relu, const float* res, int rows, int cols, int ld,
                             float* dbeta, float* dgamma, void* ws, void* stream);
int egn_bn_bwd_dz_drop_f32(const float* dy, const float* z, float p, unsigned long long seed, const int* step_dev,
                           int layer, const float* mean, const float* invstd, const float* gamma, const float* beta,
                           int relu, const float* res, const float* dbeta, const float* dgamma, float* dz,
                           float* dres, int rows, int cols, int ld, void* stream);
int egn_dropout_mask_f32(float* mask, long n, float p, unsigned long long seed, const int* step_dev, int layer,
                         void* stream);
int egn_add_f32(const float* a, const float* b, float* y, long n, void* stream);
/* *loss += weight*mean((pred-tgt)^2) (zero it first);
 * dpred (= or, with accumulate, +=) weight*2(pred-tgt)/(rows*cols).
 * weight 0.5 over all joints = the heat-map term of JointsCompositeLoss
 * (function.py:95-111), weight 1 = MSELoss1D (function.py:204-215) */
int egn_mse_f32(const float* pred, const float* tgt, int rows, int cols,
                int ld_pred, int ld_tgt, float weight, int accumulate,
                float* dpred, double* loss, void* stream);
/* The three criteria of the reference's loss_dict (libs/loss/function.py:17-20)
 * over a [rows, cols] view with row pitches; crit 0 = nn.MSELoss, 1 = nn.L1Loss,
 * 2 = nn.SmoothL1Loss (beta 1), reduction 'mean':
 *   *loss += weight * mean(c(pred - tgt)) (zero it first);
 *   dpred (= or, with accumulate, +=) weight * c'(pred - tgt) / (rows*cols).
 * Heat-map term L_hm (function.py:95-111): weight 0.5 * w_hm over all joints;
 * coordinate term L_2d (:155-168): rows 1, cols N*K*2. */
int egn_elem_loss_f32(const float* pred, const float* tgt, int rows, int cols,
                      int ld_pred, int ld_tgt, int crit, float weight,
                      int accumulate, float* dpred, double* loss, void* stream);
/* *loss += weight*mean(|pred-tgt|); dpred = weight*sign(pred-tgt)/n
 * (nn.L1Loss, the coordinate term, function.py:155-168) */
int egn_l1_f32(const float* pred, const float* tgt, long n, float weight,
               float* dpred, double* loss, void* stream);
/* cross-ratio term L_cr (function.py:113-153 calc_cross_ratio_loss + get_cr_mask,
 * img_proc.py:709-720 appro_cr): coords [N,K,2] f32, idx [L,4] int32 device
 * array of joint indices (A,B,C,D) per line (car_instance.py:83-97 'bbox12');
 * cr = |AC|^2|BD|^2 / (|BC|^2|AD|^2) / target_cr^2; a line counts when the
 * smallest non-zero distance among its four points > thres;
 * *loss += weight * sum(crit(cr,1)) / #lines kept; dcoords (may be NULL)
 * += the gradient.  crit: 0 mse, 1 l1, 2 smooth-l1.  ws: egn_cross_ratio_ws_bytes */
long egn_cross_ratio_ws_bytes(int N, int L);
int egn_cross_ratio_f32(const float* coords, int N, int K, const int* idx,
                        int L, double target_cr, float thres, int crit,
                        float weight, float* dcoords, double* loss, float* ws,
                        void* stream);
/* dz = dy*y*(1-y): backward of the coordinate head's Sigmoid (hrnet.py:461-466) */
int egn_sigmoid_bwd_f32(const float* dy, const float* y, float* dz, long n,
                        void* stream);
/* torch conv weight [Cout][Cin][KH][KW] -> packed filter of egn_conv2d_f32, on
 * the device.  dgrad 0: the forward filter.  dgrad 1: the data-gradient
 * filter (in/out channels swapped, taps rotated by 180 degrees), so that
 *   dx = egn_conv2d_f32(dy [zero-inserted when stride 2], packed, Cin' = Cout,
 *                       Cout' = Cin, stride 1, pad K-1-pad)
 * egn_packed_weight_floats = number of floats dst must hold. */
long egn_packed_weight_floats(int Cout, int Cin, int KH, int KW, int dgrad);
int egn_pack_conv_weight_f32(const float* w, int Cout, int Cin, int KH, int KW,
                             int dgrad, float* dst, void* stream);
/* Every filter of a model packed in ONE launch (the weights only change in the
 * optimizer step; a W48 training step needs ~600 packs).  descs: DEVICE array of
 *   struct { const float* w; float* dst; int32 Cout, Cin, taps, dgrad; int64 begin; }
 * (egn_pack_desc_bytes() bytes each), `begin` = running sum of the work units of the
 * preceding descriptors: egn_packed_weight_floats()/4 (one float4 each) for the direct
 * layouts (dgrad 0 / 1), egn_wino_weight_floats()/64 for the Winograd layout
 * (dgrad | 2: taps must be 9; one unit = 4 input channels x 16 frequencies),
 * egn_wino4_pack_weight_floats()/48 for the F(4x4,3x3) layout of conv_wino4.hip
 * (dgrad | 4: taps must be 9; one unit = one (output, input) channel pair). */
int egn_pack_desc_bytes(void);
int egn_pack_conv_weights_batch_f32(const void* descs_dev, int n, long total_float4,
                                    void* stream);
/* up[n][2y][2x][:] = dy[n][y][x][:], zero elsewhere; up is [N,H,W,cs] */
int egn_zero_insert2_f32(const float* dy, float* up, int N, int Ho, int Wo,
                         int H, int W, int cs, void* stream);
/* backward of one term of egn_fuse_sum_relu_f32: g [N,H>>shift,W>>shift,cs] =
 * block sums of dy*(y>0) (y NULL: no gate) */
int egn_fuse_bwd_f32(const float* dy, const float* y, float* g, int N, int H,
                     int W, int cs, int shift, void* stream);
/* Heat-map training targets for a whole batch (img_proc.py:347-409
 * generate_target): an un-normalised Gaussian dot of half-width 3*sigma centred
 * on cell int(joint/stride + 0.5) of every visible joint, clipped to the map.
 *   joints [N,K,3] f64 (x, y, unused) in input pixels; vis [N,K] f32 or NULL
 *   target [N,K,H,W] f32 out; weight [N,K] f32 out (vis, zeroed where the dot
 *   is completely out of bounds), may be NULL
 *   stride_x = input_size[0]/heatmap_size[0], stride_y = input_size[1]/heatmap_size[1] */
int egn_gaussian_targets_f32(const double* joints, const float* vis, int N, int K,
                             int H, int W, double stride_x, double stride_y,
                             double sigma, float* target, float* weight,
                             void* stream);
/* running = (1-momentum)*running + momentum*batch */
int egn_ema_f32(float* running, const float* batch, float momentum, int n,
                void* stream);
/* torch.optim.Adam (weight_decay 0, no amsgrad), step counted from 1 */
int egn_adam_step_f32(float* p, const float* g, float* m, float* v, long n,
                      float lr, float beta1, float beta2, float eps, int step,
                      void* stream);

/* the same update with the step counter (step_dev[0], incremented by the call)
 * and the learning rate (lr_dev[0]) in device memory: nothing of the iteration
 * is baked into kernel arguments, so a hipGraph captured around a whole
 * training step replays correctly */
/* [round 6] start-of-step bookkeeping in one launch: *counters[k] += 1 for the n BatchNorm num_batches_tracked words
 * (int64; `counters` = device array of their addresses) and *zero_f64 = 0 (the loss accumulator; may be NULL).
 * Replaces torch's own increment of BatchNorm.num_batches_tracked in train-mode forwards (FCmodel.py:24-43, hrnet.py:63-92). */
int egn_step_counters_i64(long long* const* counters, int n, double* zero_f64, void* stream);
int egn_adam_step_dev_f32(float* p, const float* g, float* m, float* v, long n,
                          const float* lr_dev, float beta1, float beta2,
                          float eps, int* step_dev, void* stream);
/* torch.optim.Adam(weight_decay) -- coupled L2: g += weight_decay * p before the
 * moments (libs/optimizer/optimizer.py:19-21); state as egn_adam_step_dev_f32 */
int egn_adam_l2_step_dev_f32(float* p, const float* g, float* m, float* v,
                             long n, const float* lr_dev, float beta1,
                             float beta2, float eps, float weight_decay,
                             int* step_dev, void* stream);
/* torch.optim.SGD(momentum, weight_decay), dampening 0, no Nesterov
 * (libs/optimizer/optimizer.py:23-26): g' = g + wd p; buf = g' on the first step,
 * momentum*buf + g' after; p -= lr*buf.  buf may be NULL when momentum == 0.
 * step_dev[0] is incremented first (first step <=> it becomes 1). */
int egn_sgd_step_dev_f32(float* p, const float* g, float* buf, long n,
                         const float* lr_dev, float momentum, float weight_decay,
                         int* step_dev, void* stream);

/* ------------------------------------------------------------------------
 * Programs: a recorded sequence of the launches above with every pointer
 * expressed as (slot, byte offset).  Slots are bound to base addresses before
 * a run (slot 0 = activation arena, 1 = packed weights, 2.. = user tensors),
 * so one recording serves every forward of a model at a fixed batch shape.
 * ---------------------------------------------------------------------- */
typedef struct egn_program egn_program;
typedef struct { int32_t slot; int64_t off; } egn_ref;   /* slot < 0: NULL */

egn_program* egn_program_create(int nslots);
void egn_program_destroy(egn_program* p);
int egn_program_bind(egn_program* p, int slot, void* base);
int egn_program_num_ops(const egn_program* p);

int egn_program_add_conv2d(egn_program* p, egn_ref x, egn_ref wpack,
                           egn_ref scale, egn_ref shift, egn_ref res, egn_ref y,
                           int N, int H, int W, int Cin, int cs_in,
                           int Cout, int cs_out, int KH, int KW, int stride,
                           int pad, int act, int out_nchw, int cfg);
int egn_program_add_fuse(egn_program* p, egn_ref y, int N, int H, int W, int C,
                         int cs, int nterms, const egn_ref* terms,
                         const int* shifts, int relu);
int egn_program_add_nchw_to_nhwc(egn_program* p, egn_ref x, egn_ref y, int N,
                                 int C, int H, int W, int cs);
int egn_program_add_nhwc_to_nchw(egn_program* p, egn_ref x, egn_ref y, int N,
                                 int C, int H, int W, int cs);
int egn_program_add_pixel_shuffle(egn_program* p, egn_ref x, egn_ref y, int N,
                                  int C, int H, int W, int cs, int up);
int egn_program_add_ramps(egn_program* p, egn_ref y, int N, int H, int W,
                          int cs, int c0);
int egn_program_add_decode(egn_program* p, egn_ref hm, int N, int K, int H,
                           int W, int mode, egn_ref out_xy, egn_ref out_max,
                           egn_ref out_idx);
/* Concurrency: ops added between egn_program_fork and egn_program_join run on
 * the launch lane selected by egn_program_set_lane (0 = the caller's stream,
 * 1..3 = internal side streams that start at the fork point and are waited for
 * at the join).  Ops on different lanes of one region must be independent and
 * must not share scratch buffers.  HRNet: one lane per resolution branch
 * (hrnet.py:286-287) and per fuse output (hrnet.py:291-298). */
int egn_program_fork(egn_program* p);
int egn_program_join(egn_program* p);
int egn_program_set_lane(egn_program* p, int lane);
/* tag the most recently added op (shown by the profiler); copied */
int egn_program_tag(egn_program* p, const char* tag, double flops, double bytes);

/* launch every op in order on `stream` */
int egn_program_run(egn_program* p, void* stream);
/* same, bracketing every op with hipEvents; ms[i] = duration of op i.
 * Synchronises the stream before returning. */
int egn_program_run_timed(egn_program* p, void* stream, float* ms, int n_ms);
/* capture the op sequence into a hipGraph (bindings frozen) / replay it */
int egn_program_capture(egn_program* p, void* stream);
int egn_program_replay(egn_program* p, void* stream);
/* [round 6] Ticket words of the program's K-split convolution ops (conv_wino4.hip: word 0 of an op's buffer is the error
 * word a block raises when its bounded wait for its partner runs out; 1.. are the item pairs' words, zero between runs).
 * A raised error word fails the NEXT egn_program_run / _run_timed / _replay with EGN_E_STATE (the run that raised it
 * produced invalid output); all words are zeroed again by that call.  egn_program_ticket_ops: how many ops own words.
 * egn_program_poke_ticket: TEST HOOK -- stores `value` into word `word` of K-split op `op` (0-based among those ops),
 * synchronously; the reference has no counterpart (its layers are single torch calls: libs/model/heatmapModel/hrnet.py:63-92). */
int egn_program_ticket_ops(const egn_program* p);
int egn_program_poke_ticket(egn_program* p, int op, int word, unsigned value);
/* [round 5] layer1's 1x1 pair as one op (egn_pw_pair_f32 below); w1 / shift1 / hn with slot < 0: the first product alone */
int egn_program_add_pw_pair(egn_program* p, egn_ref h, egn_ref res, egn_ref w3, egn_ref shift3,
                            egn_ref w1, egn_ref shift1, egn_ref out, egn_ref hn, int M,
                            int relu1);
/* number of kernel launches issued through egn_program_run / _run_timed / _replay
 * since the library was loaded (process wide, all devices).  Test hook: a caller
 * can prove that a forward went through this library's kernels. */
long egn_launch_count(void);
/* number of convolution-class launches issued through the DIRECT entry points (egn_conv2d_f32,
 * egn_conv2d_bnstats_f32, egn_conv2d_wgrad_f32) since the library was loaded: the training tape and the
 * torch.autograd bridge (egonet_amd/autograd.py; replaces the MIOpen convolutions torch would run for
 * libs/trainer/trainer.py:191-197) do not go through programs -- this is their proof. */
long egn_direct_conv_count(void);
/* per-op metadata for reports */
int egn_program_op_info(const egn_program* p, int i, int* kind, double* flops,
                        double* bytes, char* tag, int tag_len);

#ifdef __cplusplus
}
#endif
#endif /* EGONET_HIP_H */
