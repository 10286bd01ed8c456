// Launcher prototypes shared by the engine (engine.hip) and the kernel-level C ABI (api.hip).
#pragma once
#include <vector>

#include "common.hpp"

struct ams_comm;

namespace ams {

// ---- runtime.hip : per-device launch bookkeeping, environment knobs (read once) --------------------------
// allow `lds` bytes of dynamic LDS for kernel `fn` on the CURRENT device (no-op up to 64 KB; set once per device and size)
int func_allow_lds(const void* fn, size_t lds);
// resident blocks per CU of `fn` on the current device (cached per device / block size / LDS)
int func_blocks_per_cu(const void* fn, int threads, size_t lds, int* per_cu);
int device_cus(int* cus);
// pure bookkeeping behind func_allow_lds (CPU-testable): true the first time (device, kernel) needs a limit >= lds
bool launch_table_needs_attr(int device, const void* fn, size_t lds);
struct Knobs {                   // tuning knobs of tools/*: environment variables, read at first use, never on the launch path
    int blk_th = 0, blk_tw = 0;                  // AMS_BLK_TILE=<th>x<tw>
    int fb_walk = -1;                            // AMS_FB_WALK=<0|-2|n>: first block of the frozen path as one tile per block (0) / only the border tiles so (-2) / at most n tiles per walking block
    int fb_abl = 0;                              // AMS_FB_ABL=<bits>: measurement-only ablations of the walking first block (wrong results)
    int xwr_timed = 0;                           // AMS_XWR_TIMED=1: xdw_wreg_kernel sums per-role cycles (ams_debug_phase_cycles(2, ..))
    int blk_timed = 0;                           // AMS_BLK_TIMED=1: block_kernel sums per-phase cycles (ams_debug_phase_cycles)
    int blk_hp = -1;                             // AMS_BLK_HP=<0|1>: fp16 whole-block kernels with exact-f32 (0) / fp16 (1) project products
    char pw_force = 0; int pw_rm = 0, pw_nt = 0; // AMS_PW_FORCE=<s|l>,<RM>,<NT>
    int pw_percu = 0;                            // AMS_PW_PERCU
    bool pwx_no_tail = false;                    // AMS_PWX_NO_TAIL
    int pwx_rm = 0, pwx_nt = 0;                  // AMS_PWX_FORCE=<RM>,<NT>
    int xwr_abl = 0;                             // AMS_XWR_ABL=<bits>: the same for the weight-register streaming kernel
    int pwh_abl = 0;                             // AMS_PWH_ABL=<bits>: measurement-only ablations of the fp16 GEMM's stage loop (wrong results)
    bool pwh_set = false; int pwh_nw = 4, pwh_d = 2;                   // AMS_PWH_VARIANT=<waves per block>,<operand stages in flight>: experiment switch of the fp16 GEMM
    bool xds_set = false; int xds[6] = {0, 0, 0, 0, 0, 0};      // AMS_XDS_FORCE
    bool xwr_set = false; int xwr[5] = {0, 0, 0, 0, 0};         // AMS_XWR_FORCE
    bool wg6_eight_waves = false;                // AMS_WG6_EIGHT_WAVES: the wide tiles of the six-product weight gradient with eight waves, split 4 (k) x 2 (n)
    int wg6_split_cap = 0;                       // AMS_WG6_SPLITS: most pixel splits of the six-product weight gradient (default 32)
    int event_flags = -1;                        // AMS_EVENT_FLAGS=<hex>: flags of the stream-ordering events (default: hipEventDisableTiming)
    unsigned side_cu_mask = 0;                   // AMS_SIDE_CU_MASK=<hex word>: the fine-tune step's side stream only on the CUs whose bit is set (word repeated over the chip)
};
const Knobs& knobs();
int blk_phase_cycles(unsigned long long* h);      // k_block.hip (tools/ only)
int xwr_phase_cycles(unsigned long long* h);      // k_xdw_wreg.hip (tools/ only)
int xds_phase_cycles(unsigned long long* h);      // k_xdw_stream.hip (tools/ only)
int create_side_stream(hipStream_t* out);        // non-blocking stream for the weight gradients (runtime.hip)
int create_sync_event(hipEvent_t* out);          // event that orders streams of ONE device (runtime.hip)

// ---- comm.hip : RCCL communicator (resolved at run time) ------------------------------------------------
int comm_allreduce(ams_comm* c, void* p, size_t n, int dtype, hipStream_t st);

// ---- k_pointwise.hip : 1x1 convolutions as f32-MFMA GEMMs ---------------------------------------------
struct PwArgs {
    const float* x;        // [M, ldx]
    int64_t M;
    int K;                 // contraction length (multiple of 4)
    int Kw;                // rows of w that exist (<= K); x columns in [Kw, K) must hold finite values (they meet zeros)
    int ldx;               // row stride of x in floats (multiple of 4)
    const float* w;        // element (k, n) at w[k*w_sk + n*w_sn]
    int64_t w_sk, w_sn;
    int N;
    const float* img_bias; // [M/rows_per_img, N] added before scale/shift, or nullptr
    int64_t rows_per_img;
    const float* scale;    // [N] or nullptr
    const float* shift;    // [N] or nullptr (bias when scale == nullptr)
    int act;
    const float* res;      // [M, ldr] added after the activation, or nullptr
    int ldr;
    float* y;              // [M, ldy]
    int ldy;
    // optional second form of the result (split-bf16 kernel, vector epilogues only): bf16 parts [part][M][N], part p at
    // ysplit + p * ysplit_plane, ysplit_np = 2 | 3 parts — the operand format of the next block's expand GEMM
    uint16_t* ysplit;
    int64_t ysplit_plane;
    int ysplit_np;
    int ysplit_fmt;        // 0: bf16 parts; 1: two fp16 parts (hi | lo 2^11; ysplit_np = 2), the operand format of the fp16 streaming kernels
    // operand storage (launch_pointwise_split_f16 only): 0 = f32; 1 = "H2I", fp16 (hi | lo 2^11) pairs interleaved per 8 channels — 16 bytes
    // of hi, 16 bytes of lo — at the same 4 K bytes per row as f32 (requires ldx == K)
    int x_fmt;
    // optional column reduction fused into the epilogue (fine-tune step; plain epilogues only: no scale / shift / bias; a residual
    // operand only in mode 2 of the tiled split-bf16 kernel, where it joins the product before the mask and the sums):
    //   red_mode 1  forward BN statistics of y:  sum(y - red_center), sum((y - red_center)^2)
    //   red_mode 2  y is the gradient wrt the ACTIVATED output of a BN layer whose raw output is red_z [M, ldy]: y is multiplied by the
    //               activation's derivative act'(red_z red_scale + red_shift) before it is stored, and sum(y), sum(y xhat) are formed
    //               (xhat = (red_z - red_mean) red_rstd): the first half of that layer's BN backward
    // partial rows [rows][2][N] go to red_part, the row count to *red_rows_out (HOST pointer; 0 = the kernel chosen cannot fuse: the caller
    // runs the separate reduction)
    int red_mode;
    const float* red_center;
    const float* red_z; const float* red_scale; const float* red_shift; const float* red_mean; const float* red_rstd;
    int red_act;
    float* red_part;
    size_t red_part_floats;    // capacity of red_part: a launch whose partial rows would not fit runs WITHOUT the fused reduction (*red_rows_out = 0)
    int* red_rows_out;
    // set by the split-bf16 launcher only (mode 2 with a residual operand: the residual moves here and the plain epilogue runs)
    const float* red_res; int red_ldr;
    // optional per-element transform of the OPERAND x as it is loaded (fine-tune step: the elementwise BN passes between two layers
    // disappear into the consumer; same IEEE operations in the same order as the pass it replaces, so the product is bit-identical):
    //   x_mode 1  x' = act(x * x_v0[k] + x_v1[k])                    BN + activation of the layer that produced x (bn_act_kernel)
    //   x_mode 2  x' = x_v0[k] * x + x_v1[k] + x_v2[k] * x2[m, k]    second half of BN backward: x = gradient wrt the BN output, x2 = the
    //                                                                 layer's raw output z [M, ldx], (x_v0, x_v1, x_v2) = (A, B, C) (bn_bwd_apply_kernel, act none)
    // A kernel that cannot apply the transform materialises x' into x_tmp [M, K] (dense) first and runs on that; without x_tmp it fails.
    int x_mode; int x_act;
    const float* x_v0; const float* x_v1; const float* x_v2; const float* x2;
    float* x_tmp;
};
int launch_pointwise(const PwArgs& a, hipStream_t st);
// x' of PwArgs::x_mode written to a.x_tmp by the elementwise kernels the transform replaces; on return *b is `a` without the transform
int pointwise_materialize_x(const PwArgs& a, PwArgs* b, hipStream_t st);
bool pointwise_stream_applies(const PwArgs& a);     // the persistent streaming variant (small K x N) can take this problem
bool pointwise_transforms_on_load(const PwArgs& a);         // launch_pointwise applies a.x_mode on the operand loads (no x_tmp pass)
bool pointwise_split3_transforms_on_load(const PwArgs& a);  // the same for launch_pointwise_split3

// split-bf16 (hi + lo) late-layer variant: weights pre-split into [N][Kp] bf16 panels, Kp = K rounded up to 32
int launch_split_weights(const float* w, int64_t sk, int64_t sn, int K, int N, int Kp, uint16_t* hi, uint16_t* lo, hipStream_t st);
int launch_pointwise_split(const PwArgs& a, const uint16_t* whi, const uint16_t* wlo, int Kp, hipStream_t st);
int launch_split_weights3(const float* w, int64_t sk, int64_t sn, int K, int N, int Kp, uint16_t* hi, uint16_t* mid, uint16_t* lo,
                          hipStream_t st);
int launch_pointwise_split3(const PwArgs& a, const uint16_t* whi, const uint16_t* wmid, const uint16_t* wlo, int Kp, hipStream_t st);
// all live weight panels of a student in one launch (three-part split): job j splits w (element (k, n) at w[k*sk + n*sn]) into
// p0 | p0 + plane | p0 + 2*plane as [N][Kp]; it owns the 256-thread blocks [first_block, first_block + ceil(N*Kp / 256))
// f16 != 0: the job ALSO leaves the two fp16 parts (hi | lo 2^11) at p0 + 3 * plane | p0 + 4 * plane (forward orientation: the fine-tune
// step's forward products run on them under AMS_MATMUL_SPLIT_F16)
struct SplitJob { const float* w; int64_t sk, sn; int K, N, Kp; uint16_t* p0; int64_t plane; int64_t first_block; int f16; };
int launch_split_batch(const SplitJob* jobs_dev, int njobs, int64_t total_blocks, hipStream_t st, int write_f16);
int launch_pointwise_split1(const PwArgs& a, const uint16_t* whi, int Kp, hipStream_t st);      // one part: plain bf16 products
// two fp16 parts (k_pw_f16.hip; AMS_MATMUL_SPLIT_F16): panels [part][N][Kp] fp16, hi at whi, lo 2^11 at whi + plane; 3 MFMAs per 32 k
int launch_split_weights_f16(const float* w, int64_t sk, int64_t sn, int K, int N, int Kp, uint16_t* hi, uint16_t* lo, hipStream_t st);
bool pointwise_f16_applies(const PwArgs& a);
int launch_pointwise_split_f16(const PwArgs& a, const uint16_t* whi, int64_t plane, int Kp, hipStream_t st);
int launch_pack_h2i(const float* x, int64_t M, int C, float* out, hipStream_t st);      // f32 [M][C] -> H2I (PwArgs::x_fmt 1)
bool pointwise_split_writes_parts(const PwArgs& a);      // the split kernels will honour a.ysplit (vector epilogue)

// dw[K,N] = x[M,K]^T @ dy[M,N];  scratch holds the per-split partial products.
struct WgArgs {
    const float* x;  int ldx;  int K;
    const float* dy; int ldy;  int N;
    int64_t M;
    float* dw;             // [K, N] dense
    float* scratch; size_t scratch_floats;
    int allow_split;       // != 0: layers where the f32 MFMA kernel is matrix-pipe bound may use the 3-part bf16 kernel
    // operand transforms on load (see PwArgs::x_mode; both wgrad kernels apply them, nothing is materialised):
    //   x_mode 1   x'  = act(x * x_v0[k] + x_v1[k])
    //   dy_mode 2  dy' = dy_v0[n] * dy + dy_v1[n] + dy_v2[n] * dy2[m, n]      (dy2 [M, ldy])
    int x_mode = 0, x_act = 0; const float* x_v0 = nullptr; const float* x_v1 = nullptr;
    int dy_mode = 0; const float* dy_v0 = nullptr; const float* dy_v1 = nullptr; const float* dy_v2 = nullptr; const float* dy2 = nullptr;
};
// fused depthwise 3x3 (+BN+act) -> project 1x1 (+BN, +residual), frozen inference, split-bf16 GEMM (k_dw_project.hip)
bool dw_project_supported(int C, int N, int stride, int rate);
int launch_dw_project(const float* e, int B, int H, int W, int C, const float* w_dw, int rate, const float* sc_d, const float* sh_d,
                      int act_d, const PwArgs& p, const uint16_t* whi, const uint16_t* wlo, int Kp, hipStream_t st);
bool pointwise_wgrad_x6_applies(int64_t M, int K, int N, int ldx, int ldy);
int wgrad_x6_splits(int64_t M, int K, int N);
int launch_pointwise_wgrad_x6(const WgArgs& a, int splits, hipStream_t st);
size_t pointwise_wgrad_scratch(int64_t M, int K, int N);
int launch_pointwise_wgrad(const WgArgs& a, hipStream_t st);

// ---- k_first_block.hip : stem + depthwise + project of the first block in one kernel (frozen inference; 3 -> 32 -> 32 -> 16)
int launch_first_block(const void* frames, int dtype, int B, int H, int W, float pixel_scale, const float* w_stem,
                       const float* sc_s, const float* sh_s, int act_s, const float* w_dw, const float* sc_d, const float* sh_d,
                       int act_d, const float* w_pj, const float* sc_p, const float* sh_p, int act_p, float* y, hipStream_t st,
                       const uint16_t* w_parts = nullptr, int64_t w_plane = 0, const uint16_t* h_stem = nullptr, int64_t h_stem_plane = 0,
                       const uint16_t* h_pj = nullptr, int64_t h_pj_plane = 0);
// h_stem / h_pj: optional fp16 panels (hi | lo 2^11) of the stem [part][32][32] and of the project layer [part][16][32]: with both, the stem's and
// the project layer's products run as 3 fp16 MFMAs each (AMS_MATMUL_SPLIT_F16; takes precedence over w_parts)
// w_parts: optional three-part bf16 panels [part][32][32] of the stem weights (k = tap * 3 + channel, zero-padded 27 -> 32): with them and
// uint8 frames the stem's products run as six bf16 MFMAs, operands from a 258-entry table of the normalised byte values (f32-level)

// ---- k_ingest.hip : frame / label resize on the device (run.py:179-183)
int launch_resize_u8(const uint8_t* src, int Hs, int Ws, int C, int mode, int swap_rb, uint8_t* dst, int H, int W, hipStream_t st);

// ---- k_conv.hip : stem and depthwise ------------------------------------------------------------------
int launch_stem(const void* frames, int dtype, int B, int H, int W, const float* w, int cout, const float* scale,
                const float* shift, int act, float pixel_scale, float* y, hipStream_t st);
// im2col of the normalised, padded frame for the stem's weight gradient: out [B*Ho*Wo, 32] (27 taps + 5 zeros)
int launch_stem_im2col(const void* frames, int dtype, int B, int H, int W, float pixel_scale, float* out, hipStream_t st);
int launch_depthwise(const float* x, int B, int H, int W, int C, const float* w, int stride, int rate,
                     const float* scale, const float* shift, int act, float* y, hipStream_t st);
int launch_depthwise_dgrad(const float* dy, int B, int H, int W, int C, const float* w, int stride, int rate,
                           float* dx, hipStream_t st);
// stride-1 input gradient fused with the mask and the BN-backward sums of the layer in front (see k_conv.hip); C <= 1024
// training forward of the same layers: BN + activation of the layer in front on the tap loads, BN statistics of the result as partial rows
size_t depthwise_fwd_bn_scratch(int B, int H, int W, int C, int rate);
int launch_depthwise_fwd_bn(const float* ze, int B, int H, int W, int C, const float* w, int rate, const float* scale, const float* shift, int act,
                            const float* center, float* zd, float* scratch, int* rows_out, hipStream_t st);
size_t depthwise_dgrad_bn_scratch(int B, int H, int W, int C);
int launch_depthwise_dgrad_bn(const float* dy, int B, int H, int W, int C, const float* w, int rate, const float* z, const float* scale,
                              const float* shift, int act, const float* mean, const float* rstd, float* out, float* scratch, int* rows_out,
                              hipStream_t st);
// k_dw_train.hip: the same with the depthwise layer's own BN-backward apply pass folded in — dz_d = cA dy + cB + cC zd is formed once per
// element on the way into an LDS ring (bit-identical to bn_bwd_apply_kernel followed by the kernel above; partial rows in a different split)
size_t depthwise_dgrad_bn2_scratch(int B, int H, int W, int C, int rate);
int launch_depthwise_dgrad_bn2(const float* dy, const float* zd, const float* cA, const float* cB, const float* cC, int B, int H, int W, int C,
                               const float* w, int rate, const float* ze, const float* scale, const float* shift, int act, const float* mean,
                               const float* rstd, float* out, float* scratch, int* rows_out, hipStream_t st);
// ... and the training forward in the same tile form (a_e formed once per element on its way into LDS; z_d bit-identical to launch_depthwise_fwd_bn)
size_t depthwise_fwd_bn2_scratch(int B, int H, int W, int C, int rate);
int launch_depthwise_fwd_bn2(const float* ze, int B, int H, int W, int C, const float* w, int rate, const float* scale, const float* shift, int act,
                             const float* center, float* zd, float* scratch, int* rows_out, hipStream_t st);
size_t depthwise_wgrad_scratch(int B, int H, int W, int C, int stride, int rate);
int launch_depthwise_wgrad(const float* x, const float* dy, int B, int H, int W, int C, int stride, int rate,
                           float* dw, float* scratch, size_t scratch_floats, hipStream_t st);

// ---- k_expand_dw.hip : fused expand (1x1+BN+ReLU6) -> depthwise 3x3 (+BN+ReLU6), frozen inference ------------
bool expand_dw_supported(int Cin, int Cexp, int stride, int rate);
// stats_part != nullptr (fine-tune forward; sc_d / sh_d / act_d then describe what is WRITTEN, normally the identity): partial rows
// [*stats_rows][2][Cexp] of sum(y - center), sum((y - center)^2) of the written values, expand_dw_stats_scratch floats
size_t expand_dw_stats_scratch(int B, int H, int W, int Cexp, int stride);
int launch_expand_dw(const float* x, int B, int H, int W, int Cin, const float* w_exp, const float* sc_e, const float* sh_e, int act_e,
                     int Cexp, const float* w_dw, int stride, int rate, const float* sc_d, const float* sh_d, int act_d, float* y,
                     hipStream_t st, const float* stats_center = nullptr, float* stats_part = nullptr, int* stats_rows = nullptr);

// ---- k_xdw_train.hip : fine-tune step of the early blocks (Cin <= 32) without the 6x-expanded tensors: every consumer recomputes
// z_e = x . W_e from the block input (see the file header)
bool xdw_train_supported(int Cin, int Cexp, int stride, int rate);
size_t xdw_train_scratch(int B, int H, int W, int Cin, int Cexp);      // floats: partial rows of the passes below + one reduced row
int xdw_train_blocks(int B, int H, int W, int Cexp);
// forward statistics: partial rows  S [2][Cexp] (sum(z - center), sum((z - center)^2)) | XX [KP][KP] = x^T x | g0 [KP] = sum x
// (KP = Cin rounded up to 16) -> scratch
int launch_xdw_fwd_stats(const float* x, int B, int H, int W, int Cin, const float* w_exp, int Cexp, const float* center, float* scratch,
                         int* rows_out, int64_t* stride_out, hipStream_t st);
// backward pass 1 (given dz_d): partial rows  S [2][Cexp] | dWd [9][Cexp] | G1 [KP][Cexp]
int launch_xdw_bwd_reduce(const float* x, int B, int H, int W, int Cin, const float* w_exp, int Cexp, const float* sc_e, const float* sh_e,
                          const float* mean_e, const float* rstd_e, int act_e, const float* w_dw, int stride, const float* dz_d, float* scratch,
                          int* rows_out, int64_t* stride_out, hipStream_t st);
// first block: stem conv (as a 1x1 conv over its 27-tap patch of the frame) + first depthwise conv; partial rows
// S [2][32] | dWd [9][32] | G1 [32][32] | XX [32][32] | g0 [32]
size_t xdw_stem_scratch(int B, int fH, int fW);
int launch_xdw_bwd_reduce_stem(const void* frames, int dtype, int B, int fH, int fW, float pixel_scale, const float* w_stem, const float* sc_e,
                               const float* sh_e, const float* mean_e, const float* rstd_e, int act_e, const float* w_dw, const float* dz_d,
                               float* scratch, int* rows_out, int64_t* stride_out, hipStream_t st);
// backward pass 2: dx = (cA dy_e + cB + cC z_e) . W_e^T (+ res)
int launch_xdw_bwd_dx(const float* x, int B, int H, int W, int Cin, const float* w_exp, int Cexp, const float* sc_e, const float* sh_e,
                      int act_e, const float* w_dw, int stride, const float* dz_d, const float* cA, const float* cB, const float* cC,
                      const float* res, float* dx, hipStream_t st, const float* red_z = nullptr, const float* red_mean = nullptr,
                      const float* red_rstd = nullptr, float* red_part = nullptr, int* red_rows_out = nullptr);
// dW_e from the reduced G1 [KP][Cexp], the forward pass's (XX | g0) and the BN-backward coefficients
int launch_xdw_dwe(const float* G1, const float* xx_g0, int Cin, int Cexp, const float* w_exp, const float* cA, const float* cB, const float* cC,
                   float* dw, hipStream_t st);

// ---- k_xx_stats.hip : BN statistics of an early block's expand layer from the Gram matrix of the block input (f64 matrix pipe)
int xx_stats_blocks(int64_t M);
size_t xx_stats_scratch_doubles(int64_t M, int Cin);
int launch_xx_gram(const float* x, int64_t M, int Cin, double* scratch, double* xx64, float* xx32, hipStream_t st);
int launch_expand_stats(const double* xx64, int Cin, const float* w_exp, int Cexp, double n, const float* center, const float* gamma,
                        const float* beta, float eps, float one_minus_decay, float* moving_mean, float* moving_var, float* scale, float* shift,
                        float* save_mean, float* save_rstd, double* sums, hipStream_t st);

// ---- k_block.hip : a whole early inverted-residual block (expand -> depthwise -> project [+ input]) in one kernel ----
bool block_fused_supported(int Cin, int Cexp, int Cout, int stride, int rate, bool residual);
int launch_block_fused(const float* x, int B, int H, int W, int Cin, const float* w_exp, const float* sc_e, const float* sh_e, int act_e, int Cexp,
                       const float* w_dw, int stride, const float* sc_d, const float* sh_d, int act_d, const float* w_pj, const float* sc_p,
                       const float* sh_p, int act_p, int Cout, bool residual, float* y, hipStream_t st, const float* vecs = nullptr,
                       const uint16_t* wparts = nullptr, int64_t wplane = 0, const uint16_t* h_exp = nullptr, int64_t h_exp_plane = 0,
                       const uint16_t* h_pj = nullptr, int64_t h_pj_plane = 0, int h_pj_kp = 0);
// h_exp / h_pj: optional fp16 panels (hi | lo 2^11, `plane` apart) of the expand layer [part][Cexp][32] and the project layer
// [part][Cout][h_pj_kp]: with both, the expand and the project products run as 3 fp16 MFMAs each (AMS_MATMUL_SPLIT_F16; takes precedence over wparts)
// wparts: optional three-part bf16 panels [part][Cexp][32] of the expand weights (parts `wplane` apart): with them the expand products of
// a block with Cin 24 / 32 run as six bf16 MFMAs (f32-level) instead of eight exact-f32 MFMAs
// vecs: optional [13][Cexp] table (sc_e | sh_e | sc_d | sh_d | w_dw[9]) built once by launch_block_pack (the engine: at freeze)
int launch_first_block_tiles(const void* frames, int dtype, int B, int H, int W, float pixel_scale, const float* w_stem, const float* sc_s,
                             const float* sh_s, int act_s, const float* w_dw, const float* sc_d, const float* sh_d, int act_d, const float* w_pj,
                             const float* sc_p, const float* sh_p, int act_p, float* y, hipStream_t st, const float* vecs = nullptr);
int launch_block_pack(const float* sc_e, const float* sh_e, const float* sc_d, const float* sh_d, const float* w_dw, int Cexp, float* out, hipStream_t st);

constexpr int AMS_NP_F16 = 4;   // `np` of the streaming launchers: the two fp16 parts of split_bf16.hpp
// ---- k_xdw_stream.hip : the same fusion for the stride-16 blocks (Cin 64 / 96 / 160, stride 1, rate 1 | 2): raster-order
// streaming through an LDS ring, split-bf16 products from the expand layer's bf16 panels (np = 2 | 3 parts, `plane` apart)
bool expand_dw_stream_supported(int Cin, int Cexp, int stride, int rate);
int launch_expand_dw_stream(const float* x, const uint16_t* x_parts, int64_t x_plane, int B, int H, int W, int Cin, const float* w_f32,
                            const uint16_t* w_parts, int64_t plane, int np,
                            const float* sc_e, const float* sh_e, int act_e, int Cexp, const float* w_dw, int stride, int rate, const float* sc_d,
                            const float* sh_d, int act_d, float* y, hipStream_t st, int y_fmt = 0);

// ---- k_xdw_wreg.hip : the streaming fusion with the expand weights in registers and the operand staged in LDS (160 -> 960)
int launch_expand_dw_wreg(const uint16_t* x_parts, int64_t x_plane, int B, int H, int W, int Cin, const uint16_t* w_parts, int64_t plane,
                          int np, const float* sc_e, const float* sh_e, int act_e, int Cexp, const float* w_dw, int rate, const float* sc_d,
                          const float* sh_d, int act_d, float* y, hipStream_t st, int y_fmt = 0);
// np of the two streaming launchers: 1 .. 3 bf16 parts, or AMS_NP_F16 = the two fp16 parts of split_bf16.hpp (hi | lo 2^11; planes as for
// np = 2).  y_fmt 1 (fp16 form only): the result as fp16 pairs interleaved per 8 channels (PwArgs::x_fmt 1) instead of f32


// ---- k_elementwise.hip : BN pieces, pooling, reductions, Adam ------------------------------------------
// per-image column reductions; scratch >= image_colsum_scratch(B, C) floats
size_t image_colsum_scratch(int B, int C);
int launch_global_mean(const float* x, int B, int64_t HW, int C, float* y, float* scratch, hipStream_t st);
// column sums per image: out[b, c] = sum_hw x[b, hw, c]
int launch_image_colsum(const float* x, int B, int64_t HW, int C, int ldx, float* out, float* scratch, hipStream_t st);

// per-channel shifted sums over rows: sums[0][c] = sum(z - center[c]), sums[1][c] = sum((z-center[c])^2)  (f64)
size_t colstats_scratch(int64_t M, int C);
int launch_colstats(const float* z, int64_t M, int C, const float* center, double* sums /*[2][C]*/, float* scratch,
                    hipStream_t st);
// BN forward coefficients from the sums; updates the moving statistics when moving_mean != nullptr.
//   scale = gamma*rstd, shift = beta - mean*scale; save_mean, save_rstd for the backward pass.
int launch_bn_finalize(const double* sums, double n, int C, const float* center, const float* gamma, const float* beta,
                       float eps, float one_minus_decay, float* moving_mean, float* moving_var, float* scale,
                       float* shift, float* save_mean, float* save_rstd, hipStream_t st);
// frozen coefficients: scale = gamma*rsqrt(var+eps), shift = beta - mean*scale
int launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int C,
                   float* scale, float* shift, hipStream_t st);
// a = act(z*scale + shift) (+ res)
int launch_bn_act(const float* z, int64_t M, int C, const float* scale, const float* shift, int act, const float* res,
                  float* a, hipStream_t st);
// backward reductions: sums[0][c] = sum(dy), sums[1][c] = sum(dy * xhat), dy = da * act'(z*scale+shift)
int launch_bn_bwd_reduce(const float* da, const float* z, int64_t M, int C, const float* scale, const float* shift,
                         int act, const float* mean, const float* rstd, double* sums, float* scratch, hipStream_t st);
// per-channel (A,B,C) with dz = A*dy + B + C*z; also writes dgamma, dbeta
int launch_bn_bwd_coef(const double* sums, double n, int C, const float* gamma, const float* mean, const float* rstd,
                       float* coefA, float* coefB, float* coefC, float* dgamma, float* dbeta, hipStream_t st);
int launch_bn_param_grads(const double* sums, int C, float* dgamma, float* dbeta, hipStream_t st);
// single-GPU fusions of the above: the second reduction stage also does the per-channel arithmetic
int launch_colstats_bn(const float* z, int64_t M, int C, const float* center, double* sums, float* scratch, double n,
                       const float* gamma, const float* beta, float eps, float one_minus_decay, float* moving_mean,
                       float* moving_var, float* scale, float* shift, float* save_mean, float* save_rstd, hipStream_t st);
int launch_bn_bwd_reduce_coef(const float* da, const float* z, int64_t M, int C, const float* scale, const float* shift, int act,
                              const float* mean, const float* rstd, double* sums, float* scratch, double n, const float* gamma,
                              float* coefA, float* coefB, float* coefC, float* dgamma, float* dbeta, hipStream_t st);
int launch_bn_bwd_apply(const float* da, const float* z, int64_t M, int C, const float* scale, const float* shift, int act,
                        const float* coefA, const float* coefB, const float* coefC, float* dz, hipStream_t st);
// plain column sums over rows (bias gradient): out[c] = sum_m x[m, c]
int launch_colsum(const float* x, int64_t M, int C, int ldx, float* out, float* scratch, hipStream_t st);
int launch_adam(float* p, const float* g, float* m, float* v, const uint8_t* mask, int64_t n, float lr_t, float b1,
                float b2, float eps, hipStream_t st);
// out[i] = sum_k part[k*n + i], k ascending (deterministic second stage of split reductions)
// stride: floats between the partial rows (0 = n: dense)
// up to 32 split reductions in one launch: out[j][i] = sum_k part[j][k * stride[j] + i], i < count[j], k < splits[j]
struct ReduceJobs {
    static constexpr int kMax = 32;
    int n = 0;
    const float* part[kMax]; float* out[kMax]; int splits[kMax]; int64_t count[kMax]; int64_t stride[kMax]; int first_block[kMax];
    bool add(const float* p, int splits_, int64_t count_, float* o, int64_t stride_) {
        if (n >= kMax) return false;
        part[n] = p; splits[n] = splits_; count[n] = count_; out[n] = o; stride[n] = stride_ > 0 ? stride_ : count_; ++n;
        return true;
    }
};
int launch_reduce_batch(ReduceJobs& jobs, hipStream_t st);
int launch_reduce_splits(const float* part, int splits, int64_t n, float* out, hipStream_t st, int64_t stride = 0);
// the second (per-channel) stages of the BN reductions over partial rows [rows][2][C] produced elsewhere, row_stride floats apart
int launch_bn_fwd_finalize_partials(const float* part, int rows, int64_t row_stride, int C, double* sums, double n, const float* center,
                                    const float* gamma, const float* beta, float eps, float one_minus_decay, float* moving_mean,
                                    float* moving_var, float* scale, float* shift, float* save_mean, float* save_rstd, hipStream_t st);
int launch_bn_bwd_finalize_partials(const float* part, int rows, int64_t row_stride, int C, double* sums, double n, const float* gamma,
                                    const float* mean, const float* rstd, float* coefA, float* coefB, float* coefC, float* dgamma,
                                    float* dbeta, hipStream_t st);
int launch_partials_to_sums(const float* part, int rows, int64_t row_stride, int C, double* sums, hipStream_t st);
int launch_fill(float* p, int64_t n, float v, hipStream_t st);
int launch_round_bf16(float* p, int64_t n, hipStream_t st);       // x <- float(bf16(x)) in place (bf16-storage study)
int launch_copy(float* dst, const float* src, int64_t n, hipStream_t st);
size_t pack_fp16_scratch(int64_t n);
int launch_pack_fp16(const float* p, const uint8_t* mask, int64_t n, uint16_t* out, int64_t* n_out, int64_t* counts, hipStream_t st);

// ---- k_head.hip : fused upsample + argmax + metrics, CE gradient, phi-score confusion --------------------
// per_frame != 0: conf [B][K][K] and loss [B][2] (one confusion matrix / loss pair per frame) instead of the batch totals
int launch_upsample_argmax(const float* logits, int ld, int B, int h, int w, const int32_t* cls, int K, int H, int W,
                           const uint8_t* teacher, int NC, int32_t* labels, int64_t* conf, double* loss, hipStream_t st, int per_frame = 0,
                           int labels_u8 = 0 /* labels as uint8 [B][H][W] through the same pointer */);
int launch_ce_grad(const float* logits, int ld, int B, int h, int w, const int32_t* cls, int K, int H, int W,
                   const uint8_t* teacher, int NC, const double* loss_and_count, float* dlogits, int ldd, hipStream_t st, float empty_val = 0.f);
// loss + gradient in one pass (fine-tune step): pass 1 leaves the CE sum / valid count in loss[2] and the unnormalised gradient in
// scratch (ce_loss_grad_scratch floats); pass 2 scales by 1 / count (after its cross-rank sum) and writes dlogits [B*h*w, ldd]
bool ce_loss_grad_supported(int w, int W);
size_t ce_loss_grad_scratch(int B, int h, int w, int K);
// soft_logits != nullptr: soft-teacher targets, teacher logits [B][soft_h][soft_w][NC] f32 (utils/graph_utils.py:375-376, 403-404; k_head.hip SoftTeacher)
int launch_ce_loss_grad(const float* logits, int ld, int B, int h, int w, const int32_t* cls, int K, int H, int W, const uint8_t* teacher,
                        int NC, double* loss, float* scratch, hipStream_t st, const float* soft_logits = nullptr, int soft_h = 0, int soft_w = 0);
// empty_val: every selected class's gradient when NO pixel of the (global) batch is valid: 0, or NaN = the reference's 0 / 0
// (utils/graph_utils.py:408: loss = sum(w ce) / sum(w))
int launch_ce_combine(int B, int h, int w, const int32_t* cls, int K, int NC, const double* loss_and_count, const float* scratch,
                      float* dlogits, int ldd, hipStream_t st, float empty_val = 0.f);
// freeze-time range check of the fp16 product form: over_host[j] = 1 when any |w| of job j is > limit or not finite.  Synchronises `st`.
struct WeightRange { const float* w; int64_t n; };
int weights_beyond(const std::vector<WeightRange>& jobs, float limit, int* flags_dev /*>= jobs.size() ints + the job table*/, int* over_host, hipStream_t st);
int launch_l2_regularizer(const float* p, float* g, const uint8_t* mask, int64_t n, int n_vars, float coef, double* part_scratch /*256 doubles*/,
                          double* loss, hipStream_t st);
int launch_cross_confusion(const uint8_t* a, const uint8_t* b, int64_t n, const int32_t* lut /*[256] -> subset idx or -1*/,
                           int K, int64_t* conf, hipStream_t st);

}  // namespace ams
