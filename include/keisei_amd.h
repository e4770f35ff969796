/* keisei_amd.h -- C ABI of libkeisei_amd.so: hand-written HIP (gfx950 / CDNA4) kernels for the
 * Keisei KataGo-PPO training hot path.
 *
 * The reference (tachyon-beep/keisei) has NO native boundary on this path: every operator below is a
 * PyTorch op call site inside keisei/training/{models/se_resnet,katago_ppo,gae,value_adapter}.py
 * (SURVEY.md 2.3 K1-K23).  This header therefore defines the boundary a binding would add *underneath*
 * the reference's Python API; each entry point cites the reference statement(s) it replaces
 * (paths relative to the reference repository root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise;
 *     the library never allocates, frees or retains memory -- workspaces are caller-provided;
 *   - every launch goes to `stream` (a hipStream_t passed as void*); nothing synchronises;
 *   - return value: 0 = launched, <0 = error (KA_ERR_*), message via ka_last_error() (thread-local);
 *   - `dtype`: activation storage type of (B,81,C) NHWC tensors, KA_DTYPE_F32 (exact-fp32 MFMA / FMA,
 *     parity mode) or KA_DTYPE_BF16 (bf16 MFMA with fp32 accumulate, throughput mode);
 *   - "board" = one 9x9 position = 81 squares; activations are NHWC: element (b, p, c) at
 *     ((b*81 + p)*C + c); per-channel / per-board vectors are fp32.
 */
#ifndef KEISEI_AMD_H
#define KEISEI_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define KA_OK 0
#define KA_ERR_ARG (-1)
#define KA_ERR_HIP (-2)
#define KA_ERR_UNSUPPORTED (-3)
#define KA_DTYPE_F32 0
#define KA_DTYPE_BF16 1

/* ---- library identity / errors ------------------------------------------------------------ */
int ka_version(void);
const char* ka_target_arch(void);          /* "gfx950" */
const char* ka_last_error(void);           /* message of the last failing call on this thread */
/* Run-time switches (KA_CONV_*, KA_WGRAD_*, KA_TF_*, ...; csrc/common.h KA_OPTIONS) are read from the environment ONCE, by the
 * first launch that asks, into the table the launchers index (no getenv() on the enqueue path).  A process that changes one
 * afterwards calls ka_options_reload() (returns the number of switches set). */
int ka_options_reload(void);

/* ---- 3x3 convolution (implicit GEMM on MFMA) ----------------------------------------------------
 * Replaces nn.Conv2d(Cin, Cout, 3, padding=1, bias=False): input_conv / conv1 / conv2
 * (models/se_resnet.py:50,52,110,140) and, with a mode-1 weight pack, autograd's data gradient.
 * Optional fused input transform x' = relu?(x*in_scale[c] + in_shift[c]) + in_bias[b,c] reproduces
 * `F.relu(self.bn1(...)) + g.unsqueeze(-1).unsqueeze(-1)` (se_resnet.py:71,78) on the fly.
 * Epilogue outputs (optional): bsum[b,n] = sum over the 81 squares of the fp32 result (SE squeeze,
 * se_resnet.py:83, and the BatchNorm mean), sqpart[r,n] = per-board sum of squares
 * (r < ka_conv3x3_sqpart_rows(B)).  Requires Cin % 32 == 0 (bf16) / 16 (f32), Cout % 16 == 0. */
int ka_conv3x3_fwd(const void* in, const void* wpack, void* out, const float* in_scale, const float* in_shift,
                   const float* in_bias, int relu, float* bsum, float* sqpart, int B, int Cin, int Cout, int dtype,
                   void* stream);
/* ka_conv3x3_fwd that also writes the transformed input x' (the tensor `F.relu(self.bn1(...)) + g...` of se_resnet.py:71,78, which the
 * reference materialises and autograd saves for conv2's weight gradient) to x_out (B, 81, Cin), so that ka_conv3x3_wgrad reads it as
 * a plain operand instead of repeating the transform per tile.  x_out == NULL: ka_conv3x3_fwd.  Shapes: ka_conv3x3_fwd_keep_supported. */
int ka_conv3x3_fwd_keep_supported(int B, int Cin, int Cout, int dtype);
int ka_conv3x3_fwd_keep(const void* in, const void* wpack, void* out, const float* in_scale, const float* in_shift,
                        const float* in_bias, int relu, float* bsum, float* sqpart, void* x_out, int B, int Cin, int Cout, int dtype,
                        void* stream);
int ka_conv3x3_sqpart_rows(int B);
/* Data-gradient convolution with the surrounding BatchNorm-backward passes fused in (bf16): the input is
 * dy = in*k[0:C] + k[C:2C] + in2*k[2C:3C] (= ka_bn_bwd_apply on the fly, also written to dy_out for the weight-gradient
 * kernel); with ep_y the output is masked by the ReLU of the preceding BatchNorm, out = conv(dy)*[ep_scale*ep_y+ep_shift>0],
 * and ep_s1/ep_s2 [ka_conv3x3_sqpart_rows(B)][Cout] receive the partial sums of ka_relu_bn_bwd_reduce.  bsum = per-board
 * sums of the unmasked conv(dy) (gradient of the global-pool bias, se_resnet.py:78). */
int ka_conv3x3_dgrad_fused(const void* in, const void* in2, const float* k, void* dy_out, const void* wpack, void* out,
                           float* bsum, const void* ep_y, const float* ep_scale, const float* ep_shift,
                           const float* ep_mean, const float* ep_invstd, float* ep_s1, float* ep_s2, int B, int Cin,
                           int Cout, int dtype, void* stream);
/* ka_conv3x3_dgrad_fused with the gradient input in factored form: `du` and gate_add = [gate | add] ([2][B][Cin] fp32) with
 * dz = du * gate[b,c] + add[b,c] -- the gradient wrt bn2's output, se_resnet.py:86-90 backward: the ReLU-masked gradient of the
 * block output times the SE gate plus the squeeze path's per-board term -- as ka_block_dx_tail_bwd_du_gate leaves it.  dz is
 * formed in fp32 inside the input transform (never rounded to bf16, never stored): dy = dz*k[0:C] + k[C:2C] + in2*k[2C:3C].
 * Shapes: ka_conv3x3_dgrad_gated_supported (the two-board tower kernel: B >= 512, 256 or 128 channels, bf16). */
int ka_conv3x3_dgrad_gated_supported(int B, int Cin, int Cout, int dtype, int masked);
int ka_conv3x3_dgrad_fused_gated(const void* du, const float* gate_add, const void* in2, const float* k,
                                 void* dy_out, const void* wpack, void* out, float* bsum, const void* ep_y,
                                 const float* ep_scale, const float* ep_shift, const float* ep_mean, const float* ep_invstd,
                                 float* ep_s1, float* ep_s2, int B, int Cin, int Cout, int dtype, void* stream);
/* (Co,Ci,3,3) fp32 torch-layout weights -> MFMA B-fragment order (a derived cache; the stored parameter keeps
 * the reference's shape).  mode 0: forward, Nout = Co, Kin = Ci rounded up (zero channels); mode 1: data
 * gradient (in/out swapped, taps flipped), Nout = Ci, Kin = Co.  dst bytes = 9*(Kin/cpk)*(Nout/16)*1024. */
int ka_pack_conv3x3(const float* w, void* dst, int Co, int Ci, int Nout, int Kin, int mode, int dtype, void* stream);
/* Every layer of a network in one launch.  table = device int64 [n][8]: {src weights, dst pack, Co, Ci, Nout, Kin,
 * mode, 0} per entry, as the arguments of ka_pack_conv3x3; max_pieces = max over entries of 9*(Kin/cpk)*(Nout/16)*64. */
int ka_pack_conv3x3_multi(const long long* table, int n, long long max_pieces, int dtype, void* stream);
/* Weight gradient dW[n,c,ky,kx] = sum_{b,p} dY[b,p,n] * X'[b,p+tap,c] (autograd conv2d weight backward);
 * X' uses the same fused input transform as the forward.  slab: ka_wgrad_splits(B,Cin,Cout,target_wgs)*9*Cout*Cin floats. */
int ka_conv3x3_wgrad(const void* dy, const void* x, const float* in_scale, const float* in_shift, const float* in_bias,
                     int relu, float* slab, float* dw, int B, int Cin, int Cin_real, int Cout, int accumulate,
                     int target_wgs, int dtype, void* stream);
int ka_wgrad_splits(int B, int Cin, int Cout, int target_wgs);   /* target_wgs: 0 = 256 (one workgroup per CU) */
int ka_debug_conv_stamps(unsigned long long* stamps);   /* diagnostics only (tools/conv_stamps.py); null = off */

/* ---- layout at the model boundary ----------------------------------------------------------------
 * obs (S,Cobs,9,9) fp32 NCHW -> (B,81,Cpad) NHWC; row b is obs[idx[b]] when idx != NULL, which fuses the
 * minibatch gather `gpu_obs[idx]` (katago_ppo.py:835) into the first kernel. */
int ka_obs_to_nhwc(const float* obs, const long long* idx, void* out, int B, int Cobs, int Cpad, int dtype, void* stream);
int ka_nhwc_to_nchw(const void* in, float* out, int B, int C, int dtype, void* stream);

/* ---- BatchNorm2d, training and eval (se_resnet.py:51,53,111,121; torch defaults eps 1e-5, momentum 0.1) --
 * ka_bn_reduce: sums[0:C] = sum_b bsum[b,c], sums[C:2C] = sum_r sqpart[r,c] in fp64, fixed order
 *   (part: workspace of ka_reduce_workspace_doubles(C) doubles).  Between reduce and coeffs a caller may
 *   all-reduce `sums` across ranks (SyncBatchNorm, katago_loop.py:495-496) and pass the global element count
 *   through count_dev (device double) instead of `count`.
 * ka_bn_coeffs: scale = gamma*invstd, shift = beta - mean*scale; updates running_mean / running_var (unbiased)
 *   / num_batches_tracked exactly like nn.BatchNorm2d when the pointers are non-NULL. */
int ka_reduce_workspace_doubles(int C);
int ka_bn_reduce(const float* bsum, int B, const float* sqpart, int R, int C, double* sums, double* part, void* stream);
int ka_bn_coeffs(const double* sums, double count, const double* count_dev, const float* gamma, const float* beta,
                 float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                 float* scale, float* shift, float* mean, float* invstd, int C, void* stream);
int ka_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, float* scale, float* shift, int C, void* stream);
/* the same for n layers in one launch (eval-mode nn.BatchNorm2d at se_resnet.py:51,53,111,121): device table of n rows {gamma, beta, running_mean, running_var, scale, shift,
 * C, eps as float bits} (8 x int64 each); max_c = largest C.  Rollout inference: 82 launches -> 1. */
int ka_bn_eval_coeffs_multi(const long long* table, int n, int max_c, void* stream);
/* backward: sums = [sum dz | sum dz*yhat]; dgamma/dbeta from the LOCAL sums, dy = k[0:C]*dz + k[C:2C] + k[2C:3C]*y
 * from the (all-reduced) global sums; train = 0 gives the eval-mode derivative. */
int ka_pair_reduce(const float* p1, const float* p2, int B, int C, double* sums, double* part, void* stream);
int ka_bn_bwd_coeffs(const double* sums_local, const double* sums_global, double count, const double* count_dev,
                     const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* k,
                     int C, int train, void* stream);
/* SyncBatchNorm (katago_loop.py:495-496; torch's SyncBatchNorm all_gathers (mean, invstd, count) forward and all_reduces
 * (sum_dy, sum_dy_xmu) backward): sums[0:2C] as above from the two row sets, sums[2C] = count -- ONE vector for the caller's
 * single all-reduce per layer and direction; local_copy (optional, [2C+1]) keeps the un-reduced values (dgamma / dbeta). */
int ka_sync_reduce(const float* p1, int rows1, const float* p2, int rows2, int C, double count, double* sums,
                   double* local_copy, double* part, void* stream);
/* One launch less per BatchNorm layer when no cross-rank reduction sits between the two steps: ka_bn_reduce /
 * ka_pair_reduce with sums == NULL stop after their first stage, and these read the partials in `part` directly. */
int ka_bn_coeffs_parts(const double* part, double count, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, long long* num_batches_tracked, float momentum, float eps, float* scale,
                       float* shift, float* mean, float* invstd, int C, void* stream);
int ka_bn_bwd_coeffs_parts(const double* part, double count, const float* gamma, const float* mean, const float* invstd,
                           float* dgamma, float* dbeta, float* k, int C, int train, void* stream);
/* ... and the two steps as ONE launch (nn.BatchNorm2d's training statistics, se_resnet.py:51,53, and their autograd backward):
 * ka_bn_reduce(sums = NULL) + ka_bn_coeffs_parts, ka_pair_reduce(sums = NULL) + ka_bn_bwd_coeffs_parts.  The workgroup that
 * finishes a 64-channel column group last computes the group's coefficients from the 64 partial rows in slice order (same values
 * as the two launches, bit for bit).  counters: (C + 63) / 64 ints, zero before the first use and left zero; one such launch at
 * a time per (part, counters) pair. */
int ka_bn_reduce_coeffs(const float* bsum, int B, const float* sqpart, int R, int C, double* part, int* counters, double count,
                        const float* gamma, const float* beta, float* running_mean, float* running_var,
                        long long* num_batches_tracked, float momentum, float eps, float* scale, float* shift, float* mean,
                        float* invstd, void* stream);
int ka_pair_reduce_bwd_coeffs(const float* p1, const float* p2, int rows, int C, double* part, int* counters, double count,
                              const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* k,
                              int train, void* stream);
int ka_bn_bwd_apply(const void* dz, const void* y, const float* k, void* dy, int B, int C, int dtype, void* stream);
int ka_affine_rows(const float* in, const float* a, const float* s, float mul, float* out, int B, int C, void* stream);

/* ---- GlobalPoolBiasBlock tail and global pooling (se_resnet.py:74-77,83-98) ----------------------------
 * out = relu((scale*y+shift) * sigmoid(se[b,c]) + se[b,C+c] + res); pool[b] = [mean | max | std(correction=0) |
 * number of squares attaining the max] of `out` (4C floats; the 4th plane serves amax's tie-splitting backward).
 * se == NULL and res == NULL give relu(bn(y)) (stem, se_resnet.py:140). */
int ka_block_tail_fwd(const void* y, const float* scale, const float* shift, const float* se, const void* res, void* out,
                      float* pool, int B, int C, int dtype, void* stream);
/* ka_block_tail_fwd with the squeeze-excite FC chain of the block inside (se_resnet.py:83-86: se = se_fc2(relu(se_fc1(mean over the
 * squares of bn2(y))))): z = scale * (bsum / 81) + shift from the conv's per-board sums, h = relu(W1 z + b1), se = W2 h + b2, then the
 * tail as above with that se.  sqz_out (B,C) / se1_out (B,H) / se_out (B,2C) receive z, h and se for the backward -- the tensors
 * ka_fc_chain leaves when the chain is a launch of its own in front of ka_block_tail_fwd.  Shapes: ka_block_tail_fwd_se_supported. */
int ka_block_tail_fwd_se_supported(int C, int H, int dtype);
int ka_block_tail_fwd_se(const void* y, const float* scale, const float* shift, const float* bsum, const float* W1, const float* b1,
                         const float* W2, const float* b2, const void* res, void* out, float* pool, float* sqz_out, float* se1_out,
                         float* se_out, int B, int C, int H, int dtype, void* stream);
int ka_pool_fwd(const void* x, float* pool, int B, int C, int dtype, void* stream);
/* backward of the tail: dse = [sigmoid'(a)*sum_p du*z | sum_p du] with du = dout*[out>0]; then
 * dz = du*sigmoid(a) + dsq/81 plus the per-board BatchNorm partial sums s1p = sum dz, s2p = sum dz*yhat. */
int ka_tail_bwd_reduce(const void* dout, const void* out, const void* y, const float* scale, const float* shift,
                       const float* se, float* dse, int B, int C, int dtype, void* stream);
int ka_tail_bwd_dz(const void* dout, const void* out, const void* y, const float* se, const float* dsq, const float* mean,
                   const float* invstd, void* dz, float* s1p, float* s2p, int B, int C, int dtype, void* stream);
/* Single-pass form of the two calls above plus the squeeze-excite FC chain backward that sits between them
 * (se_resnet.py:77-86: se_fc2 -> ReLU -> se_fc1): dse (B,2C) and dh = masked gradient of the hidden layer (B,H) are
 * written for the FC weight-gradient GEMMs, dz / s1p / s2p as ka_tail_bwd_dz.  W2 = se_fc2.weight (2C,H),
 * W1 = se_fc1.weight (H,C), se1 = the saved hidden activations (B,H).  ka_tail_bwd_fused_supported() says whether
 * the shape fits the register-resident board tile. */
int ka_tail_bwd_fused_supported(int C, int H, int dtype);
int ka_tail_bwd_fused(const void* dout, const void* out, const void* y, const float* scale, const float* shift,
                      const float* se, const float* se1, const float* W2, const float* W1, const float* mean,
                      const float* invstd, void* dz, float* dse, float* dh, float* s1p, float* s2p, int B, int C, int H,
                      int dtype, void* stream);
/* da = dh*[scale*y+shift > 0] (ReLU after BatchNorm) + the same partial sums */
int ka_relu_bn_bwd_reduce(const void* dh, const void* y, const float* scale, const float* shift, const float* mean,
                          const float* invstd, void* da, float* s1p, float* s2p, int B, int C, int dtype, void* stream);
/* dx = [dxc] + [dout*[out>0]] + backward of [mean|max|std] pooling of x (ties of amax share the gradient, std
 * gradient is 0 where sigma == 0), using xpool = ka_block_tail_fwd's pool of x; dpool is (B,3C). */
int ka_block_dx(const void* dxc, const void* dout, const void* out, const void* x, const float* xpool, const float* dpool,
                void* dx, int B, int C, int dtype, void* stream);
/* The two calls that meet at a block boundary of the backward pass in ONE launch: ka_block_dx of the block above (its
 * result dx is this block's output gradient, its x this block's output: se_resnet.py:89-90, residual + ReLU) followed by
 * ka_tail_bwd_fused of this block -- x, dxc, dout_up, out_up, y are read once and dx, dz written (7 activation passes instead
 * of 9).  dx is rounded to the activation dtype before it is used: every output equals the two-launch sequence bit for bit.
 * dxc may be NULL; dout_up and out_up are both NULL where the gradient enters the tower from the heads (se_resnet.py:147-157). */
int ka_block_dx_tail_bwd_supported(int C, int H, int dtype);
int ka_block_dx_tail_bwd(const void* dxc, const void* dout_up, const void* out_up, const void* x, const float* xpool,
                         const float* dpool, void* dx, const void* y, const float* scale, const float* shift, const float* se,
                         const float* se1, const float* W2, const float* W1, const float* mean, const float* invstd, void* dz,
                         float* dse, float* dh, float* s1p, float* s2p, int B, int C, int H, int dtype, void* stream);
/* The same launch as a CHAIN over the block boundaries, one activation read shorter.  All that the residual branch one block
 * further down takes from dx is du = dx * [x > 0] (se_resnet.py:90: the ReLU after the residual sum) -- the value this launch
 * forms for its own tail -- so du_out = dx * [x > 0] is written instead of dx, and du_up (the du_out of the launch above; NULL
 * where the gradient enters from the heads) is added as it is: the block above's output is not read (4 activation reads + 2
 * writes).  dz / dse / dh / s1p / s2p equal ka_block_dx_tail_bwd's bit for bit; ka_block_dx takes a du_out as its `dout`
 * unchanged (masking twice by the same output changes nothing), which ends the chain at the first block. */
int ka_block_dx_tail_bwd_du(const void* dxc, const void* du_up, const void* x, const float* xpool, const float* dpool,
                            void* du_out, const void* y, const float* scale, const float* shift, const float* se,
                            const float* se1, const float* W2, const float* W1, const float* mean, const float* invstd, void* dz,
                            float* dse, float* dh, float* s1p, float* s2p, int B, int C, int H, int dtype, void* stream);
/* The chain launch WITHOUT dz (4 activation reads + 1 write): dz = du_out * gate_out[b,c] + add_out[b,c] (gate = sigmoid of the
 * SE gate logits, add = dsq / 81: se_resnet.py:83-90 backward) has a single reader, the conv2 data gradient, which takes the three
 * through ka_conv3x3_dgrad_fused_gated.  bf16(fmaf(du_out, gate_out, add_out)) is ka_block_dx_tail_bwd_du's dz bit for bit;
 * du_out / dse / dh / s1p / s2p are the same bits. */
int ka_block_dx_tail_bwd_du_gate(const void* dxc, const void* du_up, const void* x, const float* xpool, const float* dpool,
                                 void* du_out, const void* y, const float* scale, const float* shift, const float* se,
                                 const float* se1, const float* W2, const float* W1, const float* mean, const float* invstd,
                                 float* gate_out, float* add_out, float* dse, float* dh, float* s1p, float* s2p, int B, int C,
                                 int H, int dtype, void* stream);

/* ---- small dense layers: nn.Linear / 1x1 nn.Conv2d forward and backward (se_resnet.py:57-61,65-66,120-130) --
 * C[M,N] (+)= act(opA(A)[M,K] * opB(B)[K,N] + bias); opA(A)[m,k] = transA ? A[k*lda+m] : A[m*lda+k], likewise opB.
 * *_bf16 flags mark bf16 operands/outputs; nsplit > 1 writes raw fp32 partial slabs (reduce with ka_reduce_slabs). */
int ka_gemm(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, int lda, int ldb, int ldc,
            int transA, int transB, int a_bf16, int b_bf16, int c_bf16, int relu, int accumulate, int nsplit, void* stream);
/* Every FC weight / bias gradient of a backward pass in one launch (autograd's nn.Linear weight / bias backward for
 * se_resnet.py:57-66 global_fc / se_fc1 / se_fc2 and :125-130 value / score heads): job j computes dW_j (N,K) = dY_j^T X_j and, when db_j
 * is non-zero, db_j (N) = column sums of dY_j (dY_j (M,N) fp32; X_j M rows of ldx floats or bf16).  table: device int64
 * [njobs][10] = {dY, X, dW, db, M, N, K, ldx, x_bf16, first workgroup of the job}; a job owns ceil(N/64) *
 * ceil((K + (db != 0)) / 64) consecutive workgroups, total_wgs = their sum.  No split-K: one fixed summation order. */
int ka_gemm_grouped_wgrad(const long long* table, int njobs, int total_wgs, void* stream);
/* Two chained FC layers in one launch (se_resnet.py:57-66 global_fc / se_fc1+se_fc2, :125-130 value / score heads):
 *   y (M,N2) = W2 (N2,H) * relu(W1 (H,K1) * x' + b1) + b2,   x' = x (M rows of ldx floats, first K1 used), or with
 *   in_scale/in_shift: x'[m,k] = in_scale[k] * (x[m,k] * in_alpha) + in_shift[k] (the SE squeeze from the conv's per-board
 *   sums).  x_out (M,K1) / hidden_out (M,H) optionally keep x' / the ReLU'd hidden rows for the backward.  Exact f32
 *   matrix instructions, fp32 accumulation.  ka_fc_chain_supported() tells whether a shape is handled (K1 % 16, H in
 *   {16,32,64} or a multiple of 16 >= 128, ...); otherwise issue ka_gemm twice. */
int ka_fc_chain_supported(int K1, int ldx, int H, int N2);
int ka_fc_chain(const float* x, const float* in_scale, const float* in_shift, float in_alpha, const float* W1,
                const float* b1, const float* W2, const float* b2, float* x_out, float* hidden_out, float* y, int M, int K1,
                int ldx, int H, int N2, void* stream);
/* backward of such a chain with respect to its input, one launch: dhidden = (dy W2) * [hidden > 0] (written: the dY of W1's
 * weight gradient), dx = dhidden W1; W2T (H, N2) and W1T (K1, H) are transposed copies of the nn.Linear weights kept current
 * by ka_transpose_multi (table rows {src, dst, rows, cols}).  Replaces autograd's input gradients of global_fc
 * (se_resnet.py:62-63, 71). */
int ka_fc_chain_bwd(const float* dy, const float* hidden, const float* W2T, const float* W1T, float* dhidden_out, float* dx,
                    int M, int N2, int H, int K1, void* stream);
int ka_transpose_multi(const void* table, int n, int max_tiles, void* stream);
int ka_reduce_slabs(const float* slab, float* out, int nsplit, long long n, int accumulate, void* stream);
/* two slab sets of one split count in one launch (a layer's weight- and bias-gradient partials); same sums, same order */
int ka_reduce_slabs2(const float* slab_a, float* out_a, long long na, const float* slab_b, float* out_b, long long nb, int nsplit,
                     void* stream);
int ka_colsum(const float* A, const float* Bm, float* part, float* part2, int M, int N, int nsplit, void* stream);
int ka_relu_mask(float* g, const float* h, long long n, void* stream);
/* policy-head BatchNorm over fp32 rows (M = B*81, N = policy_channels; se_resnet.py:121,144) */
int ka_rows_affine_relu(const float* in, const float* scale, const float* shift, float* out, long long M, int N, void* stream);
int ka_rows_sq_sums(const float* A, float* part, float* part2, int M, int N, int nsplit, void* stream);
int ka_rows_bn_sums(const float* da, const float* in, const float* mean, const float* invstd, float* part, float* part2,
                    int M, int N, int nsplit, void* stream);
int ka_rows_bn_bwd(float* dr, const float* in, const float* p0, const float* p1, long long M, int N, int mode, void* stream);

/* ---- fused KataGo-PPO minibatch loss (katago_ppo.py:857-924, :33-57; value_adapter.py:98-126) -----------
 * ka_policy_loss: masked log-softmax over A = 11259 actions, log-prob gather, clipped surrogate, entropy over
 *   legal actions -- per-sample terms AND dL/dlogits in one pass over the logits.  Per-sample inputs (legal,
 *   actions, old_lp, adv) are rows of the epoch dataset addressed through idx (NULL = identity).
 *   w_policy = lambda_policy/B, w_entropy = entropy_coeff/B; gscale = optional device loss scale (GradScaler).
 *   flags[0] |= NaN in raw logits, flags[1] |= 1: a sample without legal action (the reference's two guards,
 *   katago_ppo.py:861-871), flags[1] |= 2: an action id outside [0, A) (the reference's gather traps on the device).
 *   legal: bool rows (S,A) when legal_words == 0, else packed rows (S,legal_words) uint32 with bit j of word w = action
 *   32 w + j and legal_words == ka_mask_words(A) -- the device rollout store's column (ka_rollout_append).
 * ka_value_loss: W/D/L cross-entropy (ignore_index -1, all-ignored -> 0), score MSE, their gradients, the mean
 *   reductions of the per-sample policy terms, and the value metrics of compute_value_metrics (katago_ppo.py:60-78).
 *   out[9] = {policy_loss, value_ce, score_mse, entropy, total, n_valid, value_accuracy, frac_win, frac_draw};
 *   acc[4] += {policy, value (combined lambda-weighted when combined_value_metric), score, entropy}. */
int ka_policy_loss(const float* logits, const void* legal, const long long* actions, const float* old_lp, const float* adv,
                   const long long* idx, float* dlogits, float* new_lp, float* rowloss, float* rowent, int* flags,
                   const float* gscale, float clip_eps, float w_policy, float w_entropy, int B, int A, int legal_words,
                   void* stream);
/* Rollout action selection (katago_ppo.py:566-584): probs[b] = softmax of logits[b] over the legal actions, 0 elsewhere;
 * nlegal[b] = number of legal actions (0 -> the caller raises the reference's error); flags[0] |= NaN in the logits.
 * legal as in ka_policy_loss (bool rows, or packed rows with legal_words = ka_mask_words(A)). */
int ka_masked_softmax(const float* logits, const void* legal, float* probs, int* nlegal, int* flags, int B, int A,
                      int legal_words, void* stream);
/* ka_policy_sample: the whole tail of select_actions in one launch (katago_ppo.py:567-612 masked_fill / softmax /
 * Categorical.sample / log_prob / zero-legal guard, :536-541 scalar value, value_adapter.py:56-65 blend).  logits (B,A) fp32 or
 * bf16 (logits_bf16 != 0); legal as in ka_masked_softmax; seed: any 64-bit value, one uniform per (seed, row) by a 64-bit mix;
 * actions (B) int64, logp (B) = log softmax_masked(logits)[action], nlegal (B); values (B) optional: P(W)-P(L) of vlogits (B,3),
 * blended with clamp(score,-1,1) by alpha when score != NULL.  flags[0] |= NaN in the logits, flags[1] |= a row with no legal action. */
int ka_policy_sample(const void* logits, int logits_bf16, const void* legal, int legal_words, long long seed,
                     const float* vlogits, const float* score, float alpha, long long* actions, float* logp, float* values,
                     int* nlegal, int* flags, int B, int A, void* stream);
/* Supervised policy cross-entropy (keisei/sl/trainer.py:150-152): rowloss[b] = logsumexp(logits[b]) - logits[b][t],
 * t = targets[idx ? idx[b] : b]; dlogits (optional) = w_policy * (softmax - onehot) [* *gscale].  flags[0] |= NaN logits,
 * flags[1] |= target outside [0,A).  ka_value_loss then supplies the W/D/L cross-entropy, the score MSE and the means
 * (rowent = zeros, entropy_coeff = 0). */
int ka_policy_ce(const float* logits, const long long* targets, const long long* idx, float* dlogits, float* rowloss,
                 int* flags, const float* gscale, float w_policy, int B, int A, void* stream);
int ka_value_loss(const float* vlogits, const float* score, const long long* cats, const float* targets,
                  const long long* idx, const float* rowloss, const float* rowent, float* dvlogits, float* dscore,
                  float* out, float* acc, const float* gscale, float lambda_policy, float lambda_value, float lambda_score,
                  float entropy_coeff, int combined_value_metric, int B, void* stream);
/* P(W) - P(L), optionally blended with clamp(score,-1,1) (katago_ppo.py:533-541, value_adapter.py:76-96) */
int ka_scalar_value(const float* vlogits, const float* score, float alpha, float* out, int B, void* stream);

/* ---- GradScaler.unscale_ + clip_grad_norm_ + torch.optim.Adam.step + GradScaler.update (katago_ppo.py:926-933) --
 * tab: nt records {float* p, const float* g, float* m, float* v, long long n}; blk_tensor / blk_off map each of the
 * nblocks chunks (ka_adam_chunk() elements) to (record, element offset).  ctl[0] = unscaled global grad norm,
 * ctl[2] = 1 when the step was vetoed (inf/NaN gradients or a guard flag); step_state[0] = applied steps;
 * scaler = {scale, growth_tracker} or NULL. */
int ka_adam_chunk(void);
int ka_clip_adam_step(const void* tab, const int* blk_tensor, const long long* blk_off, int nblocks, double* partial,
                      float* ctl, float* step_state, float* scaler, const int* guard_flags, float* acc_gnorm,
                      float max_norm, float lr, float beta1, float beta2, float eps, void* stream);

/* ---- Generalised Advantage Estimation (gae.py:8-296) and advantage normalisation (katago_ppo.py:797-798) --
 * (T,N) grids, one launch; override: NaN = default bootstrap; lengths (N) selects the padded variants;
 * f64 != 0: rewards/values/next_value/override/adv are double.  Bit-identical to compute_gae_gpu's op order. */
int ka_gae(const void* rewards, const void* values, const float* term, const void* next_value, const void* override_,
           const long long* lengths, void* adv, int T, int N, double gamma, double lam, int f64, void* stream);
int ka_normalize_advantages(const float* x, float* out, long long n, void* stream);

/* ---- Device-resident rollout store (KataGoRolloutBuffer.add / flatten, katago_ppo.py:128-388) --
 * ka_rollout_append: one timestep of n transitions.  Sources are the tensors add() receives (obs fp32 (n,obs_elems),
 *   legal bool (n,A), actions / cats / env_ids int64, log_probs / values / rewards / score / override fp32, dones /
 *   terminated bytes); destinations are the store's columns already offset to the first row written; the mask row
 *   is packed to ka_mask_words(A) uint32 (bit j of word w = action 32 w + j).  env_ids / override / d_env_ids /
 *   d_override may be NULL; a store with an override column and no override supplied gets NaN (= no override).
 *   The reference's input guards (katago_ppo.py:244-266) come back as flags: [0] terminated without done,
 *   [1] category outside {-1,0,1,2}, [2] NaN score target, [3] = float bits of max |score target| seen.
 * ka_unpack_mask_bits: bool rows out[r] = packed row idx[r] (idx NULL = identity) -- flatten()'s legal_masks.
 * ka_pack_mask_bits: packed rows from bool rows. */
int ka_mask_words(int A);
int ka_rollout_append(const float* obs, const void* legal, const long long* actions, const float* log_probs,
                      const float* values, const float* rewards, const void* dones, const void* terminated,
                      const long long* cats, const float* score, const long long* env_ids, const float* override_,
                      float* d_obs, void* d_bits, long long* d_actions, float* d_log_probs, float* d_values, float* d_rewards,
                      void* d_dones, void* d_terminated, long long* d_cats, float* d_score, long long* d_env_ids,
                      float* d_override, int* flags, int n, int obs_elems, int A, void* stream);
/* as ka_rollout_append with the legal masks already PACKED: legal_bits (n, ka_mask_words(A)) uint32 rows (the device env's
 * StepResult.legal_mask_bits, PendingTransitions.finalize()["legal_mask_bits"]) are copied word for word. */
int ka_rollout_append_packed(const float* obs, const void* legal_bits, const long long* actions, const float* log_probs,
                             const float* values, const float* rewards, const void* dones, const void* terminated,
                             const long long* cats, const float* score, const long long* env_ids, const float* override_,
                             float* d_obs, void* d_bits, long long* d_actions, float* d_log_probs, float* d_values, float* d_rewards,
                             void* d_dones, void* d_terminated, long long* d_cats, float* d_score, long long* d_env_ids,
                             float* d_override, int* flags, int n, int obs_elems, int A, void* stream);
int ka_unpack_mask_bits(const void* bits, const long long* idx, void* out, int rows, int A, void* stream);
int ka_pack_mask_bits(const void* legal, void* bits, int rows, int A, void* stream);

/* ---- Pending learner transitions of the split-merge rollout (PendingTransitions.create / accumulate_reward / finalize,
 * katago_loop.py:139-250).  Slots = one row per game: obs fp32 (n, obs_elems), legal masks PACKED (n,
 * ka_mask_words(A)) uint32, actions int64, log_probs / values / rewards / score fp32, valid bytes.
 * ka_pending_open (create(), katago_loop.py:172-200): the games with env_mask != 0 take this step's rows; masks come as bool
 *   rows (s_legal, (n, A)) and are packed on the way, or as packed rows (s_bits) and are copied.  flags[0] = 1 and nothing is
 *   written when a selected slot still holds a transition (the reference's RuntimeError).
 * ka_pending_accumulate (accumulate_reward(), :203-211): rewards[valid] += add[valid].
 * ka_pending_settle (finalize(), :213-250): rows of the games with fin_mask & valid, in game order, into the o_* columns
 *   (n rows allocated; flags[1] = rows written); dones / terminated are this step's flags for all n games, floats
 *   (flags_are_f32 != 0) or bytes; add_rewards (may be NULL) is accumulated first, as accumulate_reward() would; o_cats =
 *   the value-head labels of _compute_value_cats (:75-92) for the settled rows; settled slots are released (valid_out, a
 *   second buffer: the launch still counts over `valid`) and their rewards zeroed. */
int ka_pending_open(float* obs, void* bits, long long* actions, float* log_probs, float* values, float* rewards, float* score,
                    void* valid, const void* env_mask, const float* s_obs, const void* s_legal, const void* s_bits,
                    const long long* s_actions, const float* s_log_probs, const float* s_values, const float* s_rewards,
                    const float* s_score, int* flags, int n, int obs_elems, int A, void* stream);
int ka_pending_accumulate(float* rewards, const void* valid, const float* add_rewards, int n, void* stream);
int ka_pending_settle(float* obs, void* bits, long long* actions, float* log_probs, float* values, float* rewards, float* score,
                      const void* valid, void* valid_out, const void* fin_mask, const void* dones, const void* terminated,
                      int flags_are_f32, const float* add_rewards, float* o_obs, void* o_bits, long long* o_actions,
                      float* o_log_probs, float* o_values, float* o_rewards, float* o_dones, float* o_terminated, float* o_score,
                      long long* o_env_ids, long long* o_cats, int* flags, int n, int obs_elems, int A, void* stream);

/* ---- the eval-mode residual tower in one launch (rollout inference, SURVEY 8 f2: katago_ppo.py:543-617 calling
 * se_resnet.py:67-75 for every block under no_grad / eval()).  One workgroup carries one board through all blocks; the
 * activations live in LDS.  x_in / x_out (B, 81, C) bf16, pool_in / pool_out (B, 4C) fp32 [mean|max|std|-]; blocks = device
 * table of nblocks rows of 14 pointers: {conv1 pack, conv2 pack (ka_pack_conv3x3 mode 0), bn1 scale, bn1 shift, bn2 scale,
 * bn2 shift (eval), global_fc[0].weight (G,3C), .bias, global_fc[2].weight (C,G), .bias, se_fc1.weight (R,C), .bias,
 * se_fc2.weight (2C,R), .bias}. */
int ka_tower_eval_supported(int C, int G, int R, int dtype);
int ka_tower_eval(const void* x_in, const float* pool_in, void* x_out, float* pool_out, const void* blocks, int nblocks, int B,
                  int C, int G, int R, int dtype, void* stream);

/* ---- the vectorised shogi environment on the device (shogi-engine/crates/shogi-gym/src/vec_env.rs:556-855
 * VecEnv(num_envs, max_ply, "katago", "spatial"), with the rules of shogi-core/src/{movegen,attack,rules,game}.rs, the
 * observation planes of shogi-gym/src/katago_observation.rs:41-92 + observation.rs:81-153 and the action indices of
 * spatial_action_mapper.rs:138-279).  One wave per game; all buffers are device memory owned by the caller:
 *   state  n x ka_shogi_env_state_bytes() bytes: board[81] (piece.rs:10-19 bytes) hands[2][7] side in_check - ply key reps
 *   keys   n x max(max_ply,1) u64, checks n x max(max_ply,1) u8: position key / "mover stood in check" of every ply
 *   obs_mode 1 = "katago" 50 planes, 0 = "default" 46 planes (observation.rs:1-15); action_mode 1 = "spatial" A = 11 259,
 *   0 = "default" A = 81*80*2 + 81*7 = 13 527 (action_mapper.rs:17-110); ka_shogi_env_action_space(mode) = A.
 *   obs (n,planes,9,9) fp32; mask (n,A) bool bytes and/or mask_bits (n,ceil(A/32)) u32 (bit j of word w = action 32w+j;
 *   at least one of the two); current_players (n) u8.
 * ka_shogi_env_reset: VecEnv::reset (vec_env.rs:617-645) -- or, with refresh != 0, derive key / check / masks from the
 *   board, hands and side the caller has written into `state` (test fixtures; ply and history start at 0).
 * ka_shogi_env_step: VecEnv::step (vec_env.rs:651-700, apply_moves :340-460).  Phase 1 checks every action against
 *   prev_mask / prev_mask_bits (the masks of the previous call); `err` points at FOUR ints = two 64-bit words, 8-byte aligned:
 *   word 0 (cleared by every call) = ((n - i) << 32) | (uint32) action for the first refused env i, and then NO game moves (the
 *   reference raises before mutating); word 1 latches the first non-zero word 0 and is never cleared by the library: the
 *   caller zeroes it when it has reported the refusal, so a flag read late survives any number of further steps.  Phase 2: make_move, check_termination (game.rs:355-387: move
 *   limit, fourfold repetition / perpetual check, 24-point impasse, no legal move), rewards for the mover
 *   (vec_env.rs:98-124), captured hand-type (255 none), TerminationReason, ply, material balance, episode counters
 *   stats[4] u64 {completed, drawn, truncated, total ply}; finished games write terminal_obs (other rows are left as they
 *   were) and restart from the start position; then observation and masks of every game's position to move. */
int ka_shogi_env_state_bytes(void);
int ka_shogi_env_action_space(int action_mode);
int ka_shogi_env_reset(void* state, void* keys, void* checks, int n, int max_ply, int obs_mode, int action_mode, float* obs,
                       void* mask, void* mask_bits, void* current_players, int refresh, void* stream);
int ka_shogi_env_step(void* state, void* keys, void* checks, const long long* actions, int n, int max_ply, int obs_mode,
                      int action_mode, const void* prev_mask, const void* prev_mask_bits, int* err, float* obs, void* mask, void* mask_bits,
                      float* rewards, void* terminated, void* truncated, float* terminal_obs, void* current_players,
                      void* captured, void* term_reason, void* ply_count, int* material, void* stats, void* stream);

/* ---- transformer encoder path (BASELINE config 5; keisei/training/models/transformer.py:37-95: nn.Linear(50, d),
 * row/col nn.Embedding, nn.TransformerEncoder(nn.TransformerEncoderLayer(d, nhead, 4d, batch_first, norm_first), L),
 * nn.Linear(81 d, 11259), value head).  Tokens are (B*81, d) row-major, bf16 (autocast) or fp32 (parity mode; its linear
 * layers run on ka_gemm).  Dropout masks (nn.Dropout p = 0.1 inside the encoder layer, transformer.py:45-50) come from a
 * counter-based hash of (seed, element index): recomputed in the backward, never stored. */
/* C = epilogue(A * B^T), bf16 operands [M][lda] / [N][ldb] (K % 32 == 0), fp32 accumulate: forward nn.Linear (B = bf16
 * weight copy), its input gradient (B = transposed copy) and its weight gradient (A, B = transposed activations, nsplit
 * slabs over the token axis).  epilogue = +bias, ReLU, dropout, +residual, bf16 or fp32 store. */
int ka_tf_gemm_nt(const void* A, const void* B, void* C, const float* bias, const void* residual, int M, int N, int K,
                  int lda, int ldb, int ldc, int c_bf16, int relu, int nsplit, float drop_p, unsigned long long seed,
                  void* stream);
int ka_tf_gemm_nt_slabs(int K, int nsplit);
/* C (bf16) = (A B^T) * dropout_keep(seed, element) * [relu_act > 0]: the input-gradient GEMM of linear2 carrying the
 * backward of the dropout and ReLU that follow linear1 in the forward (transformer.py:45-50 via nn.TransformerEncoderLayer) */
int ka_tf_gemm_nt_masked(const void* A, const void* B, void* C, const void* relu_act, int M, int N, int K, int lda, int ldb,
                         int ldc, float drop_p, unsigned long long seed, void* stream);
/* dW = dY^T X straight from the row-major activations (weight gradient of nn.Linear, transformer.py:40-61): C[z][N][ldc]
 * fp32 slabs over token ranges (z < ka_tf_gemm_tn_slabs(M, nsplit); one slab = the result), A [M][lda] (N columns) and
 * B [M][ldb] (K columns) bf16, N / K / lda / ldb multiples of 8.  Operand fragments by LDS transpose reads: no transposed copies. */
int ka_tf_gemm_tn(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int nsplit, void* stream);
/* the same launch also producing the bias gradient of the layer: colsum [ka_tf_gemm_tn_slabs(M, nsplit)][N] = column sums of A
 * per token range (four more MFMAs per step against an all-ones fragment; db = sum of the slabs) */
int ka_tf_gemm_tn_bias(const void* A, const void* B, float* C, float* colsum, int M, int N, int K, int lda, int ldb, int ldc,
                       int nsplit, void* stream);
int ka_tf_gemm_tn_slabs(int M, int nsplit);
int ka_tf_transpose_pad(const void* in, void* out, int M, int N, int ldi, int ldo, int dtype, void* stream);
int ka_tf_cast_pad(const void* in, void* out, long long M, int N, int ldi, int ldo, int dtype, void* stream);
/* the bf16 operand copies of n nn.Linear weights (transformer.py:37-52: in_proj / out_proj / linear1 / linear2 of every encoder
 * layer, policy_fc) in one launch and from one read of each weight: table rows {W fp32 (N, K), out (N, ldo) bf16, outT (K, ldt) bf16,
 * N, K, ldo, ldt, first tile}, ldo % 8 == 0, ldt % 8 == 0; total_tiles = sum of ceil(ldt / 64) * ceil(ldo / 64).
 * Same values as ka_tf_cast_pad + ka_tf_transpose_pad per layer. */
int ka_tf_weights16_multi(const void* table, int n, int total_tiles, void* stream);
/* x[b,s,:] += row_embed[s/9] + col_embed[s%9] (transformer.py:84-87) and the embedding gradients (scratch: 65*81*d floats) */
int ka_tf_add_pos(void* x, const float* row_embed, const float* col_embed, int B, int d, int dtype, void* stream);
int ka_tf_pos_grad(const void* dx, float* scratch, float* drow, float* dcol, int B, int d, int dtype, void* stream);
/* nn.LayerNorm(d) (eps 1e-5) forward / backward; part: (ka_tf_layernorm_parts(M) + 1) * 2 * d floats; dx = LN'(dy) + dres */
int ka_tf_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, long long M,
                        int d, float eps, int dtype, void* stream);
int ka_tf_layernorm_parts(long long M);
int ka_tf_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                        const void* dres, void* dx, float* part, float* dgamma, float* dbeta, long long M, int d, int dtype,
                        void* stream);
/* the same with a second output dx_drop = dx * dropout_keep(seed, element) (NULL: none): the gradient entering the sub-layer
 * below through its dropout, written in the same pass */
int ka_tf_layernorm_bwd_drop(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                             const void* dres, void* dx, void* dx_drop, float drop_p, unsigned long long seed, float* part,
                             float* dgamma, float* dbeta, long long M, int d, int dtype, void* stream);
/* out = in * keep [* (act > 0)] [+ res] */
int ka_tf_drop_apply(const void* g_in, const void* act, const void* res, void* g_out, long long n, float drop_p,
                     unsigned long long seed, int dtype, void* stream);
int ka_tf_colsum(const void* a, float* part, float* out, long long M, int N, int nsplit, int dtype, void* stream);
int ka_tf_mean_pool(const void* x, float* pooled, int B, int d, int dtype, void* stream);
int ka_tf_head_grad(const float* dpooled, const void* dflat, void* dx, int B, int d, int dtype, void* stream);
int ka_tf_tanh(float* v, long long n, void* stream);
int ka_tf_tanh_bwd(const float* dy, const float* y, float* dx, long long n, void* stream);
/* nn.MultiheadAttention(d, nhead, batch_first) core over the 81 squares: softmax(Q K^T / sqrt(dh)) V per (board, head) on
 * the matrix cores, dropout on the probabilities in training; qkv [B*81][3d] (in_proj output), out [B*81][d],
 * lse [B][H][81] for the backward.  dh <= 64. */
int ka_tf_attention_fwd(const void* qkv, void* out, float* lse, int B, int H, int dh, float drop_p, unsigned long long seed,
                        int dtype, void* stream);
int ka_tf_attention_bwd(const void* qkv, const void* dout, const float* lse, void* dqkv, int B, int H, int dh, float drop_p,
                        unsigned long long seed, int dtype, void* stream);
/* ... with the forward's output handed in: D[query] = rowsum(dout * out) makes dQ and dK / dV two independent launches on the
 * register-resident path (bf16, dh <= 32); other shapes fall through to ka_tf_attention_bwd */
int ka_tf_attention_bwd_o(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int H, int dh,
                          float drop_p, unsigned long long seed, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KEISEI_AMD_H */
