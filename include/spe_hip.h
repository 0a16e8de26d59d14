/*
 * spe_hip.h - C ABI of libspe_hip.so: the MI355X (gfx950) kernels behind the SPE hot path.
 *
 * The reference (MingXiangL/SPE) has no native code: every "kernel" on its hot path is an
 * implicit ATen/cuBLAS launch issued from Python.  Each entry point
 * below therefore cites the reference *Python* site whose device work it replaces.  The
 * Python host code in spe_amd/ binds these with ctypes (spe_amd/lib.py); INTEGRATION.md shows
 * the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (the kernels never allocate);
 *   - tensors are fp32, row-major, densely packed unless a leading dimension is passed;
 *   - `stream` is the hipStream_t the work is enqueued on; calls are asynchronous;
 *   - return value: 0 on success, a positive hipError_t on launch failure, a negative value
 *     for an unsupported argument combination; no exception crosses the boundary;
 *   - dropout masks are a pure function of (seed, offset, element index) - Philox4x32 with 7 rounds (csrc/common.h) -
 *     so forward and backward regenerate the same mask (the attention's backward loads the 1-bit keep flags its forward stored).
 */
#ifndef SPE_HIP_H
#define SPE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* spe_stream_t; /* == hipStream_t */

/* 2 since round 2 (changed signatures: spe_hungarian, spe_adamw_flat, spe_layernorm_fwd, spe_attn_contract,
 * spe_talking_fused_plan; new entry points); 3 since round 3 (fp16 forward formats of the fused attention: spe_attn_contract
 * takes `fmt`, spe_attn_pack_multi kinds carry an element-format bit, spe_talking_fused expects fp16 Q / K fragments and writes
 * fp16 P'd blocks; split-operand GEMM entry points); 4: deterministic reductions - spe_set_reduce_workspace is new and required
 * before any entry point that sums across workgroups, spe_box_loss takes L, spe_linear_small_fwd / _bwd are new; 5 (round 4):
 * the flash-style talking-heads entry points spe_talking_flash_* are new; 6 (round 5): spe_talking_bwdq_* are new; 7 (round 6): ONE attention
 * backward composition - spe_talking_fused(_bits / _plan), spe_attn_merge, spe_talking_flash_rows, spe_talking_flash_dv, spe_talking_bwdq_pass1 removed,
 * spe_talking_stats(_plan) new (the statistics pass alone), spe_rowdot new, spe_layernorm_res_bwd takes dy2 */
int spe_abi_version(void);

/* ---- reduction workspace --------------------------------------------------------------------
 * Every sum across workgroups on the gradient path (bias / LayerNorm gamma, beta / LayerScale gamma column sums of
 * spe_layernorm_bwd, spe_layernorm_res_bwd, spe_colsum, spe_cvt_bf16, spe_gemm_bf16nt_ex, spe_layerscale_residual_bwd(16), the
 * loss sums of spe_focal_loss; reference: the reductions autograd performs for models/cait.py:376-416, models/transformer.py:
 * 279-287 and models/conditional_detr.py:253-275) is taken in a FIXED order - per-workgroup partials in slabs, integer tickets, the
 * last arriver adds (csrc/det_reduce.h) - so gradients are bitwise reproducible run to run; no fp32 atomics.  The slabs and tickets
 * live in caller-owned device memory registered here once per process (256-B aligned, >= 2 MiB; 16 MiB covers every shape of
 * the model path); entry points that need it return -4 when none is registered or it is too small.  All launches that use it
 * must be ordered on one stream at a time.  ws = NULL unregisters. */
int spe_set_reduce_workspace(void* ws, size_t bytes, spe_stream_t stream);
/* Deferred sums (round 4).  The fixed-order tree costs the last workgroup of every launch six dependent trips to the coherence point
 * (5-8 us; ~250 launches per step).  When the destination of a sum is a parameter gradient inside the caller's all-reduce buckets, nothing
 * reads it before the bucket is reduced / the optimiser runs: the owner of the buckets registers their address ranges
 * (spe_reduce_defer_ranges: n <= 64 ranges, n = 0 switches deferral off) and an arena (spe_reduce_defer_arena: caller-owned device memory,
 * 256-B aligned, a few tens of MB); entry points whose destinations ALL lie inside the ranges (spe_layernorm_bwd / _res_bwd,
 * spe_layerscale_residual_bwd16(d), spe_cvt_bf16 with colsum, spe_gemm_bf16nt_ex(d) with colsum, spe_colsum_bf16_blocks) then only leave
 * their per-workgroup partial rows in the arena, and spe_reduce_flush adds the rows of ALL pending launches to their destinations in ONE
 * kernel (members in index order: bitwise reproducible), stream-ordered after them.  The caller MUST flush before anything reads those
 * gradients (before a bucket's all-reduce, at the end of the backward); the library flushes by itself when its table or the arena is full.
 * spe_reduce_pending: launches waiting for a flush.  -3: ranges / arena changed while sums are pending. */
int spe_reduce_defer_ranges(const void* const* ptrs, const size_t* bytes, int n);
int spe_reduce_defer_arena(void* arena, size_t bytes);
int spe_reduce_flush(spe_stream_t stream);
int spe_reduce_pending(void);

/* ---- contraction ----------------------------------------------------------------------
 * C[z] = act(alpha * opA(A[z]) @ opB(B[z]) + bias) for z = (z0, z1) in batch0 x batch1, each
 * operand addressed as base + z0*s?0 + z1*s?1.  transA=0: A is [M,K] (lda); transA=1: A is
 * [K,M].  transB=0: B is [K,N] (ldb); transB=1: B is [N,K] (nn.Linear weight layout).
 * act: 0 none, 1 ReLU, 2 exact-erf GELU; C2 (optional) receives the pre-activation.
 * splitk>1: K is split over workgroups and atomically accumulated into a PRE-ZEROED C
 * (bias/act/C2 must be null/0).  splitk<-1: |splitk| splits, split z stores its partial result into the
 * private slab C + z*S (no atomics); the caller sums the slabs (spe_colsum over |splitk| rows of S floats), S = M*ldc
 * for a single matrix, else the extent of the batched C: ((batch0-1)*sC0 + (batch1-1)*sC1 + (M-1)*ldc + N) rounded up to 4.  precision: 0 = bf16 MFMA operands, fp32 accumulate;
 * 1 = 3-term bf16 split (~fp32 accuracy).
 * Replaces nn.Linear / torch.bmm / `@` at reference models/cait.py:376-390 (qkv, QK^T, PV, proj),
 * :114-133 (class attention), timm Mlp fc1/fc2 (cait.py:409), Conv2d patch embed (cait.py:526),
 * models/transformer.py:368-425 (decoder projections, FFN), models/attention.py:353,375,378,
 * models/conditional_detr.py:104-110 (heads), and every one of their autograd backward GEMMs. */
int spe_gemm_f32(const float* A, const float* B, float* C, const float* bias, float* C2,
                 int M, int N, int K, long lda, long ldb, long ldc, int transA, int transB,
                 int batch0, int batch1, long sA0, long sA1, long sB0, long sB1, long sC0, long sC1,
                 float alpha, int act, int splitk, int precision, spe_stream_t stream);

/* Same contraction with the A operand stored as bf16 when a_bf16 = 1 (lda and sA0/sA1 count bf16
 * elements): consumes the transposed score tensors written by spe_talking_fused for the PV / dV /
 * dQ / dK products of reference models/cait.py:389 and its autograd. */
int spe_gemm_ex(const void* A, int a_bf16, const float* B, float* C, const float* bias, float* C2,
                int M, int N, int K, long lda, long ldb, long ldc, int transA, int transB,
                int batch0, int batch1, long sA0, long sA1, long sB0, long sB1, long sC0, long sC1,
                float alpha, int act, int splitk, int precision, spe_stream_t stream);

/* Tile edge (128 or 64) spe_gemm_* will use for an (M, N, batch) problem; callers use it to choose split-K. */
int spe_gemm_tile(int M, int N, int nbatch);

/* ---- LayerNorm (nn.LayerNorm; reference models/cait.py:403,407 eps 1e-6,
 * models/transformer.py:264-265,342-344 eps 1e-5).  C % 4 == 0, C <= 1024.
 * bwd: dgamma/dbeta are ACCUMULATED into (pre-zeroed or running) buffers; add (optional, [R][C]; may alias dx) is added to
 * dx - the gradient arriving over the residual path around the normalised branch (x feeds both: cait.py:404-405), which
 * autograd would otherwise sum with one more elementwise launch.
 * fwd: y16 (optional, bf16 [R][C]): the same result rounded to bf16 - the operand of the Linear that consumes y, written by
 * the same pass instead of by a separate spe_cvt_bf16 launch; y16lo (optional, needs y16): bf16(y - bf16(y)), the low part of
 * the SPLIT operand of precision mode bf16s (see spe_gemm_bf16nt). */
int spe_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                      float* rstd, long R, int C, float eps, void* y16, void* y16lo, spe_stream_t stream);
/* spe_layernorm_fwd_h (round 5): the same, with the second 16-bit copy yh16 = IEEE fp16(y) (saturating) instead of the low part - the
 * operand of a single-term fp16 forward product (the backbone MLP's fc1 in precision mode bf16s); y16 stays the backward's bf16 operand. */
int spe_layernorm_fwd_h(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                        float* rstd, long R, int C, float eps, void* y16, void* yh16, spe_stream_t stream);
int spe_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                      const float* rstd, float* dx, float* dgamma, float* dbeta, long R, int C,
                      const float* add, spe_stream_t stream);
/* spe_layernorm_bwd_ls (round 5): spe_layernorm_bwd plus, in the same pass, the LayerScale backward of the node that CONSUMES dx - in a
 * backbone block the input of a norm is out = res + ls_gamma * ls_y, the output of the previous branch's Linear (reference models/cait.py:
 * 404-405), so dx is that node's `dout`: ls_dy16 [R][C] = bf16(ls_gamma * dx) (the operand of that Linear's backward GEMMs), ls_db[c] += sum_r
 * ls_gamma[c] dx[r][c] (its bias gradient), ls_dg[c] += sum_r dx[r][c] ls_y[r][c] (the LayerScale gradient) - what
 * spe_layerscale_residual_bwd16 computes from dx in a launch of its own.  ls_y fp32 [R][C]; no dropout / DropPath in that branch; C <= 512.
 * The caller must know that dx has no other consumer (the sums land in the gradients before the consuming node runs). */
int spe_layernorm_bwd_ls(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                         float* dgamma, float* dbeta, long R, int C, const float* add, const float* ls_y, const float* ls_gamma,
                         void* ls_dy16, float* ls_db, float* ls_dg, spe_stream_t stream);

/* ---- bf16-operand Linear GEMM (benchmark precision mode): C = act(alpha * A16 B16^T + bias), both operands
 * k-contiguous bf16 (lda, ldb, K multiples of 8; 16-B aligned bases), fp32 C / C2 (pre-activation) / bias.
 * splitk < 0: |splitk| K-slices write private slabs C + z*M*ldc (no bias/act; sum them with spe_colsum).
 * The three products of nn.Linear and its autograd (reference models/cait.py:376,390,409;
 * models/transformer.py:368-425) are NT products of copies made by spe_cvt_bf16:
 *   y = x W^T: (x16, W16) ; dx = dy W: (dy16, W16T) ; dW = dy^T x: (dy16T, x16T), contraction zero padded.
 * spe_cvt_bf16: out[R][ldo] = bf16(x) (round to nearest even, the rounding spe_gemm_f32 applies while staging)
 * and/or outT[C][ldt] = transpose, columns R..ldt-1 zero filled; colsum[c] += sum_r x[r][c] (fp32; the bias gradient
 * of the Linear, from the same read).  Any of the three outputs may be NULL.  aux != NULL: x is first multiplied by the
 * activation derivative at aux (act 1: ReLU with aux = forward output, 2: erf-GELU with aux = pre-activation; same
 * layout and leading dimension as x) - the backward of a fused Linear+activation without an fp32 intermediate.
 * SPLIT operands (precision mode "bf16s", the forward products): A16lo / B16lo (both or neither; same leading dimensions as
 * A16 / B16) hold bf16(x - bf16(x)) of the fp32 operands, written next to the high parts by every producer (out_lo of
 * spe_cvt_bf16 / spe_cvt_bf16_multi, y16lo of spe_layernorm_fwd, out16lo of spe_attn_contract and spe_gemm_bf16nt_ex); the
 * product is then A_hi B_hi + A_lo B_hi + A_hi B_lo - ~16 significant operand bits instead of 8 at 3x the MFMA work of kernels
 * that are load / store bound (no split-K with split operands).  * act bits 8 / 9 (round 4, the decoder's memory-side projections - north_star's "decoder cross-attention GEMM", reference
 * models/transformer.py:389-396): bit 8 = A16 / B16 hold IEEE fp16 and the product is single-term on v_mfma_f32_16x16x32_f16 (no lo
 * parts); bit 9 = C is an IEEE fp16 [M][ldc] matrix (saturating) - what spe_attn_pack_multi's fp16-source jobs consume.  Both need
 * splitk = 1, no C2, ldc % 4 == 0; bit 8 needs M >= 2048 and K % 64 == 0.  spe_cvt_f16: fp32 [R, C] -> IEEE fp16 (C, ldx, ldo % 4 == 0). */
int spe_cvt_f16(const float* x, long ldx, int R, int C, void* out, long ldo, spe_stream_t stream);
int spe_gemm_bf16nt(const void* A16, const void* B16, const void* A16lo, const void* B16lo, float* C, const float* bias,
                    float* C2, int M, int N, int K, long lda, long ldb, long ldc, float alpha, int act, int splitk,
                    spe_stream_t stream);
/* nn.Linear on a few hundred rows (the decoder / encoder / head side: reference models/transformer.py:21-33, 206-250, 355-427,
 * models/conditional_detr.py:68-116) as ONE launch each way - these products are bound by dependent launches, not by throughput.
 * spe_linear_small_fwd: y[R][ldc] = act(x W^T + bias) from the fp32 activations x [R][ldx] and the cached bf16 weight W16 [N][K]
 *   (W16lo != NULL: split operands, x is split into (hi, lo) while it is staged - precision mode bf16s); pre (optional): the
 *   pre-activation; x16_out (optional): bf16(x) [R][K], all the backward needs of x.  act: 0 none, 1 ReLU, 2 exact-erf GELU.
 * spe_linear_small_bwd: dx [R][K] = dy' W, dW [N][K] = dy'^T x, db [N] = colsum(dy') in one grid, dy' = dy (fp32 [R][N]) times
 *   the derivative of the fused activation at aux (act 1: forward output, act 2: pre-activation; aux NULL: none); x16 [R][K] and
 *   WT16 [K][N] bf16; any of dx / dW / db may be NULL.  dW and db are OVERWRITTEN (no zeroing needed), db is summed in a fixed
 *   order (no atomics).  K, N multiples of 8, operands 16-B aligned; -2 otherwise. */
int spe_linear_small_fwd(const float* x, long ldx, const void* W16, const void* W16lo, const float* bias, float* y, float* pre,
                         void* x16_out, int R, int N, int K, long ldc, int act, spe_stream_t stream);
int spe_linear_small_bwd(const float* dy, const float* aux, int act, const void* x16, const void* WT16, float* dx, float* dW,
                         float* db, int R, int N, int K, spe_stream_t stream);
/* Group form (round 4): nblk Linears of equal shape [N][K] applied to ONE input - the decoder layer's query-side projections
 * (reference models/transformer.py:368-372: sa_qcontent / sa_kcontent / sa_v of tgt; 369-371, 399: sa_qpos / sa_kpos of every layer and
 * ca_qpos of the first on query_pos) - as one launch each way; every Linear keeps its own cached weight copies, output and gradient
 * buffers: the arguments are HOST arrays of nblk device pointers, nothing is stacked.  nblk <= 16.
 * spe_linear_small_group_fwd: y[i] [R][N] = x W_i^T + bias[i] (+ add[i], a contiguous fp32 [R][N]; `add` or single elements may be NULL - the
 *   decoder's q = sa_qcontent_proj(tgt) + sa_qpos_proj(query_pos), transformer.py:373-374, without an add launch) (W16lo != NULL: split operands, every
 *   element non-NULL); N % 32 == 0.
 * spe_linear_small_group_bwd: dx [R][K] = sum_i dy[i] W_i (WT16[i] = bf16 W_i^T [K][N]), dW[i] [N][K] = dy[i]^T x16, db[i] [N] =
 *   colsum(dy[i]), overwritten, fixed summation order; dy[i] == NULL: output i received no gradient - it adds nothing to dx and its
 *   dW[i] / db[i] are not written.  dx, dW, db (or single elements of dW / db) may be NULL.  N % 128 == 0. */
int spe_linear_small_group_fwd(const float* x, long ldx, const void* const* W16, const void* const* W16lo, const float* const* bias,
                               const float* const* add, float* const* y, void* x16_out, int R, int nblk, int N, int K, spe_stream_t stream);
int spe_linear_small_group_bwd(const float* const* dy, const void* x16, const void* const* WT16, float* dx, float* const* dW,
                               float* const* db, int R, int nblk, int N, int K, spe_stream_t stream);
/* spe_gemm_bf16tn: C[m][n] = alpha * sum_r A16[r][m] * B16[r][n] - the weight gradient dW = dy^T x of a Linear (autograd of
 * reference models/cait.py:376,390,409, models/transformer.py:368-425) on ROW-MAJOR bf16 operands A16 [R, lda] (M columns)
 * and B16 [R, ldb] (N columns): the contraction runs over rows, the MFMA operands are formed by LDS transpose reads
 * (ds_read_b64_tr_b16), no transposed copies of x or dy exist.  splitk < 0: |splitk| private slabs of M*ldc floats over
 * the row range (sum them with spe_colsum).  Operands 16-B aligned; lda, ldb, M, N multiples of 8. */
int spe_gemm_bf16tn(const void* A16, const void* B16, float* C, int M, int N, int R, long lda, long ldb, long ldc,
                    float alpha, int splitk, spe_stream_t stream);
/* spe_gemm_bf16nt_ex: the same product with an epilogue that feeds the NEXT GEMMs directly (no fp32 round trip, no
 * separate conversion launch):  v = alpha A16 B16^T + bias ; C2 = v (optional) ; v = act(v), or with aux != NULL
 * v = v * act'(aux) (aux [M][ldc]: act 1 = ReLU with the forward output, act 2 = erf-GELU with the pre-activation) ;
 * C = v (optional fp32) ; out16[M][ld16] = bf16(v) ; out16T[N][ld16t] = bf16(v)^T with columns M..ld16t-1 zero
 * (ld16t <= M rounded up to 64) ; colsum[n] += sum_m v.  Any output may be NULL.  res/rgamma != NULL (act 0, no aux):
 * the LayerScale residual of the block is applied by the epilogue, C = res[m][n] + rgamma[n] * v (res [M][ldc]), while C2
 * keeps v for the gamma gradient.  half_flags bit 0: C2 is stored as IEEE fp16 [M][ldc] (saturating) instead of fp32; bit 1: aux
 * holds IEEE fp16 [M][ldc] - the saved pre-activation of the MLP, of which only gelu'(.) is ever taken.  Used by the fused MLP of the
 * backbone block (reference models/cait.py:405-416 = timm Mlp fc1 -> GELU -> fc2 inside x + gamma_2 * mlp(norm2(x)), and its autograd).
 * half_flags bits 2 / 3 (round 5: that MLP's FORWARD products in precision mode bf16s; profiles/r05_error_budget.jsonl - fp16 operands in
 * fc1 + fc2 alone cost <= 2.6e-4 on every weighted loss key at cfg2 / cfg5 full depth): bit 2 = A16 / B16 hold IEEE fp16, the product is
 * single-term on v_mfma_f32_16x16x32_f16 (no low parts; M >= 2048, K % 64 == 0) ; bit 3 (needs bit 2) = out16lo receives IEEE fp16(v)
 * (saturating) - the operand of the NEXT fp16 product - next to out16 = bf16(v), the backward's operand. */
int spe_gemm_bf16nt_ex(const void* A16, const void* B16, const void* A16lo, const void* B16lo, float* C, const float* bias,
                       float* C2, void* out16, void* out16lo, long ld16, void* out16T, long ld16t, float* colsum, const float* aux,
                       const float* res, const float* rgamma, int M, int N, int K, long lda, long ldb, long ldc, float alpha, int act,
                       int half_flags, spe_stream_t stream);
/* spe_gemm_bf16nt_exd (round 4): spe_gemm_bf16nt_ex with the backbone block's training rates inside the epilogue (reference models/cait.py:
 * 390-391 proj_drop, timm Mlp drop after GELU and after fc2, :404-416 drop_path): after the activation (or its derivative) v is multiplied
 * by the dropout keep scale of element m * N + n - the (seed, offset) stream of spe_dropout on the row-major [M][N] result, N % 4 == 0 - and
 * with res the epilogue writes C = res + sample_scale[m / rows_per_sample] * rgamma * v (sample_scale NULL: 1).  C2 keeps the RAW v: the
 * backward (spe_layerscale_residual_bwd16d) applies the same mask and scale. */
int spe_gemm_bf16nt_exd(const void* A16, const void* B16, const void* A16lo, const void* B16lo, float* C, const float* bias,
                        float* C2, void* out16, void* out16lo, long ld16, void* out16T, long ld16t, float* colsum, const float* aux,
                        const float* res, const float* rgamma, int M, int N, int K, long lda, long ldb, long ldc, float alpha, int act,
                        int half_flags, float p_drop, uint64_t seed, uint64_t offset, const float* sample_scale, long rows_per_sample,
                        spe_stream_t stream);
int spe_cvt_bf16(const float* x, long ldx, int R, int C, void* out, void* out_lo, long ldo, void* outT, long ldt, float* colsum,
                 const float* aux, int act, spe_stream_t stream);
/* spe_cvt_bf16_h (round 5): out = bf16(x), out_h = IEEE fp16(x) (saturating; same layout as out), outT as above - one pass. */
int spe_cvt_bf16_h(const float* x, long ldx, int R, int C, void* out, void* out_h, long ldo, void* outT, long ldt, spe_stream_t stream);
/* spe_cvt_bf16_multi: the row-major and transposed bf16 copies of njobs contiguous fp32 matrices in ONE launch (every
 * Linear weight after an optimizer step).  jobs_dev: device array of njobs records of 64 bytes
 *   { const float* x; void* out; void* outT; long ldt; int R, C, tile0, tiles_c; void* out_lo; long flags; }     out, out_lo [R][C], outT [C][ldt], ldt >= R;
 * tiles_c = ceil(C/64), tile0 = running sum of tiles_c * ceil(max(R, ldt)/64) over the preceding jobs; total_tiles = that
 * sum over all jobs.  out, out_lo or outT may be NULL per job.  flags bit 0: out_lo receives IEEE fp16(x) instead of the low part. */
int spe_cvt_bf16_multi(const void* jobs_dev, int njobs, int total_tiles, spe_stream_t stream);

/* ---- masked softmax over scores[B,H,Nq,ld] (Nk valid columns per row).
 * mask[B,Nk] (1 = padded key, -inf) or null; P = softmax; Pd = dropout(P) written only when
 * p_drop > 0.  Reference models/attention.py:363-373, models/cait.py:125-131 (class attention).
 * bwd: dS = P*(dP - sum dP*P) with dP = dPd*keepscale; dS may alias dPd. */
int spe_softmax_fwd(const float* S, const unsigned char* mask, float* P, float* Pd, int B, int H, int Nq,
                    int Nk, long ld, float p_drop, uint64_t seed, uint64_t offset, spe_stream_t stream);
int spe_softmax_bwd(const float* dPd, const float* P, float* dS, int B, int H, int Nq, int Nk, long ld,
                    float p_drop, uint64_t seed, uint64_t offset, spe_stream_t stream);

/* ---- talking-heads score transform (reference models/cait.py:379-387):
 * S' = proj_l(S) over heads, P = softmax_k(S'), P' = proj_w(P), Pd = attn_drop(P').
 * H in {4,6,8}.  P may alias S.  bwd consumes dPd, saved P and re-computed raw scores S and
 * writes dS (may alias dPd) plus `nblocks` rows of partial [dWl | dbl | dWw | dbw] sums
 * (row length 2*(H*H+H)) into ws; the caller column-sums ws (spe_colsum).  nblocks <= B*Nq. */
int spe_talking_softmax_fwd(const float* S, const float* Wl, const float* bl, const float* Ww, const float* bw,
                            float* P, float* Pd, int B, int H, int Nq, int Nk, long ld, float p_drop,
                            uint64_t seed, uint64_t offset, spe_stream_t stream);
int spe_talking_softmax_bwd(const float* dPd, const float* P, const float* S, const float* Wl, const float* Ww,
                            float* dS, float* ws, int nblocks, int B, int H, int Nq, int Nk, long ld,
                            float p_drop, uint64_t seed, uint64_t offset, spe_stream_t stream);

/* ---- fused talking-heads attention (reference models/cait.py:377-389 + autograd): qkv -> softmax(proj_l(scale q k^T)) -> proj_w -> attn_drop -> @ v
 * with NO N x N tensor in HBM for the forward (the backward's dS is the only one, bf16).  ONE composition:
 *   forward : spe_attn_pack_multi -> spe_talking_stats -> spe_attn_merge_rows -> spe_talking_flash_fwd
 *   backward: spe_attn_pack_multi(dO) -> spe_talking_bwdk_pass1 (D, dWw, dbw, dV) -> spe_talking_bwdq_pass2 (dS, dWl, dbl, dQ)
 *             -> spe_talking_wgrad_reduce -> spe_attn_contract(trans = 1) (dK)
 * (rounds 1-5 carried four backward compositions behind a developer switch - spe_talking_fused modes 1-3, spe_talking_fused_bits, spe_attn_merge,
 * spe_talking_flash_rows, spe_talking_flash_dv, spe_talking_bwdq_pass1; ABI 7 removed them: profiles/HISTORY_r05.md.)
 *
 * Operand format.  spe_attn_pack / spe_attn_pack_multi write 16-bit MFMA operand fragment records, per (b, h, 16-row tile):
 *   FULL steps of [lane][8] = scale * x[b, tile*16 + (lane&15), h, st*32 + (lane>>4)*8 + i], then - when dh % 32 is in
 *   1..16 - one 16-wide tail step of [lane][4] = scale * x[.., FULL*32 + (lane>>4)*4 + i]   (dh = 48: 1.5 KB/record)
 * from x[b][n][h][d] (element strides sb, sn, sh).  Qf and Kf are FP16 records (spe_attn_pack_multi kinds 0 + 16; Qf packed with scale * log2(e):
 * the kernels work in the log2 domain), Vf and dOf BF16 (kind 0): the forward quantities are O(1) and take the 3 extra mantissa bits, gradients
 * keep bf16's range.  Blocked score tensors (dS): bf16 16 x 16 blocks [B,H,nt,nt][64][4], nt = ceil(N/16), lane l of block (qt,kt) = query
 * qt*16+(l&15), keys kt*16+4*(l>>4)+i.
 *
 * spe_talking_stats: partial softmax statistics (running max, sum of exp2) of S' = proj_l(scale q k^T) per (b, g, q) -> ws_stats
 *   (B*nt*8*H*32 floats); spe_talking_stats_plan: the launcher's work split for an `nwg` workgroup budget (workgroups walk (q-tile pair, key tile)
 *   steps) - steps per workgroup (the value spe_attn_merge_rows needs) and workgroups used.
 * spe_attn_merge_rows: merges ws_stats into M (log2-domain row max), IL (1 / row sum) [B,H,N] and the row constants rows [B][Np][H] =
 *   bl[g] log2(e) - M + log2(IL), zero for the rows N .. Np-1 (the addend that turns Wl S into log2 P).  Np: multiple of 16, >= 16 ceil(N / 16).
 * spe_talking_wgrad_reduce: column sums of ws_w (nwg rows of 2*(H*H+H) = [dWl|dbl|dWw|dbw]) STORED (not accumulated) into the four gradient
 *   buffers in fixed order; a NULL destination is skipped.
 * Supported: H in {4,8}, head dim <= 64 and the LDS bound below.  Returns -2 otherwise (use the materialised path: spe_talking_softmax_*). */
int spe_attn_pack(const float* x, long sb, long sn, long sh, int B, int N, int H, int dh, float scale,
                  void* out, spe_stream_t stream);
int spe_talking_stats_plan(int B, int N, int nwg, int* steps_per_wg, int* nwg_used);
int spe_talking_stats(const void* Qf, const void* Kf, const float* Wl, const float* bl, float* ws_stats,
                      int B, int H, int N, int dh, int nwg, spe_stream_t stream);
int spe_attn_merge_rows(const float* ws, float* M, float* IL, const float* bl, float* rows, int Np, int B, int H, int N,
                        int steps_per_wg, spe_stream_t stream);
int spe_talking_wgrad_reduce(const float* ws_w, int nwg, int H, float* dWl, float* dbl, float* dWw, float* dbw,
                             spe_stream_t stream);

/* ---- flash forward: P' goes from the head mix straight into the matrix instructions that consume it, the result accumulates in registers.
 * spe_talking_flash_plan: the launcher's flattened work split for an `nwg` workgroup budget (a workgroup keeps 8 tiles of 16 rows
 *   resident and streams the tiles of the other axis): steps per workgroup, workgroups launched, major tile groups per image, and
 *   Np = the padded row count of the row-constant arrays.  The partial-result workspace `ws` holds
 *   B * nmajor * 8 (slots) * 128 * H * 16 * ceil(dh / 16) floats.
 * spe_talking_flash_fwd: O[b, q, g*dh + d] = sum_key P'd[b,g][q,key] v[b, key, g, d] with P'd = attn_drop(proj_w(softmax(proj_l(
 *   scale q k^T)))); Qf / Kf fp16 fragments (spe_attn_pack_multi kind 0 + 16, Qf packed with scale * log2(e)), V16 fp16 (kind
 *   1 + 16), c0 = the row constants of spe_attn_merge_rows.  O16 / O16lo (optional): bf16(O) and bf16(O - bf16(O)), same addressing - the
 *   operand of the output projection.  keepbits (p_drop > 0 and a backward will follow): uint32 [B][nt][nt][64] - the dropout keep flags of
 *   every 16 x 16 tile (bit hp * 8 + 2 r + e of lane l: query l & 15, key 4 (l >> 4) + r, head 2 hp + e); the backward kernels LOAD them
 *   (one dword per lane and tile) instead of regenerating the masks.
 * Supported: H in {4, 8}, 13 * H * ceil(dh / 16) * 512 + 3072 bytes of LDS <= 160 KB; -2 otherwise. */
int spe_talking_flash_plan(int B, int N, int nwg, int* steps_per_wg, int* nwg_used, int* nmajor, int* rows_padded);
int spe_talking_flash_fwd(const void* Qf, const void* Kf, const void* V16, const float* Wl, const float* Ww, const float* bw,
                          const float* c0, int Np, float* ws, float* O, void* O16, void* O16lo, void* keepbits, int B, int H, int N, int dh,
                          int nwg, float p_drop, uint64_t seed, uint64_t offset, spe_stream_t stream);

/* ---- the two backward kernels (reference: the autograd of models/cait.py:377-389 - proj_l, softmax, proj_w, attn_drop; csrc/attn_flash_bwd.hip).
 * A workgroup of 4 waves, ONE wave per SIMD (512 registers each): a wave keeps its tile's fragment records and 96 accumulators in AccVGPRs and
 * streams the tiles of the other axis through LDS.
 * spe_talking_bwdq_plan: the work split for an `nwg` workgroup budget - steps per workgroup, workgroups launched, major (4-tile) groups per
 *   image.  Workspaces: ws_q / ws_v B * nmajor * 8 * 4 * H * ceil(dh / 16) * 256 floats, ws_d B * nmajor * Np * H floats, ws_w
 *   (nwg_used * 4) rows of 2 * (H * H + H) floats - row layout [dWl | dbl | dWw | dbw], pass 2 fills the first half, pass 1 the second;
 *   spe_talking_wgrad_reduce(ws_w, 4 * nwg_used, ...) sums them.
 * spe_talking_bwdk_pass1 (KEY-major: a wave keeps one key tile's K / V records and 96 dV accumulators and streams the q-tiles): S, S', P recomputed
 *   ONCE for Drows[b][q][h'] = sum_key dP[h'] P[h'] ([B][Np][H], rows >= N zero; D summed over the 4 key tiles of a major in ws_d), the dWw / dbw
 *   partials, and dv[b, key, h, :] = sum_q P'd[b,h][q,key] dO[b, q, h, :] (P'd = dropout(proj_w(P)); element strides ob, on, oh; dv fp32 and /
 *   or dv16 bf16, either may be NULL).  Qf, Kf fp16 records (the forward's), dOf, Vf bf16 records (kind 0), dO16 = bf16 dO in the 16-wide layout
 *   (kind 1), c0 [B][Np][H] of spe_attn_merge_rows, Np >= 16 ceil(N / 16) + 64.
 * spe_talking_bwdq_pass2 (QUERY-major): dS = proj_l^T (P (dP - D)) as bf16 16 x 16 blocks (the dK contraction reads them), dWl / dbl partials,
 *   and dq[b, q, h, :] = scale * sum_key dS[b,h][q,key] k[b, key, h, :] accumulated in registers (element strides ob, on, oh; dq fp32 and / or
 *   dq16 bf16 with the same addressing, either may be NULL).  K16 = bf16 k in the 16-wide layout (kind 1).
 * keepbits: the dropout keep flags of spe_talking_flash_fwd, required when p_drop > 0.  Supported: H in {4, 8}, head dim <= 64 within the LDS
 * bound (H = 8: head dim <= 48); -2 otherwise. */
int spe_talking_bwdk_pass1(const void* Qf, const void* dOf, const void* dO16, const void* Kf, const void* Vf, const float* Wl, const float* Ww,
                           const float* bw, const float* c0, int Np, float* ws_d, float* ws_v, float* ws_w, float* Drows, float* dv, void* dv16,
                           long ob, long on, long oh, const void* keepbits, int B, int H, int N, int dh, int nwg, float p_drop, spe_stream_t stream);
int spe_talking_bwdq_plan(int B, int N, int nwg, int* steps_per_wg, int* nwg_used, int* nmajor);
int spe_talking_bwdq_pass2(const void* Qf, const void* dOf, const void* Kf, const void* Vf, const void* K16, const float* Wl, const float* Ww,
                           const float* c0, const float* Drows, int Np, float* ws_q, float* ws_w, void* dS, float* dq, void* dq16,
                           long ob, long on, long oh, float scale, const void* keepbits, int B, int H, int N, int dh, int nwg, float p_drop,
                           spe_stream_t stream);

/* ---- streaming contractions of a blocked 16-bit score tensor T (dS of spe_talking_bwdq_pass2):
 *   trans = 0: out[b, q, h, :]   = alpha * sum_key T[b,h][q,key] x[b, key, h, :]   (`attn @ v`, cait.py:388; dQ)
 *   trans = 1: out[b, key, h, :] = alpha * sum_q   T[b,h][q,key] x[b, q, h, :]     (dV, dK of the same autograd)
 * X16 = spe_attn_pack16(x): bf16 [B,H,nt,ceil(dh/16)][64][4] = x[b, tile*16+4*(lane>>4)+i, h, dtile*16+(lane&15)]
 * (x element strides sb, sn, sh; 0 outside).  out element strides ob (batch), on (row), oh (head), unit d stride.
 * fmt: 0 = bf16 T x bf16 X16 (dQ, dK); 1 = fp16 T x fp16 X16, trans = 0 only (O = P'd V: pass alpha = 2^-8 for mode 1's scale);
 * 2 = fp16 T x bf16 X16, trans = 1 only (dV = P'd^T dO).
 * Head dim <= 64, returns -2 otherwise.  ws / counters (optional, both or neither): ws_floats floats of scratch and zero-
 * initialised 32-bit counters (at least B*H*2; left zero again by every launch) that let the launcher cut the groups of
 * output tiles beyond the last full round of resident workgroups into quarters of the contraction range, combined by the
 * last arriver in fixed order - without them every group is one workgroup.  One workspace serves all launches of a stream. */
int spe_attn_pack16(const float* x, long sb, long sn, long sh, int B, int N, int H, int dh, void* out, spe_stream_t stream);
/* njobs <= 6 packs of [B,Ns[i],H,dhs[i]] views in one launch: job i reads xs[i] (element strides strides[3i..3i+2] =
 * batch, row, head), multiplies by scales[i] and writes the spe_attn_pack (kinds[i] = 0), spe_attn_pack16 (1) or
 * full-32-steps-only (2: ceil(dh/32) steps of [lane][8], no tail step; spe_mha_*) layout to outs[i]; kinds[i] + 16: the same
 * layout with IEEE fp16 elements (saturating at +-65504) instead of bf16.  The pointer/stride tables are HOST arrays (copied
 * into the kernel arguments). */
int spe_attn_pack_multi(int njobs, const float* const* xs, const long* strides, const float* scales, const int* kinds,
                        void* const* outs, const int* Ns, const int* dhs, int B, int H, spe_stream_t stream);
int spe_attn_contract(const void* T, const void* X16, float* out, long ob, long on, long oh, int B, int H, int N, int dh,
                      int trans, int fmt, float alpha, float* ws, unsigned int* counters, long ws_floats, void* out16,
                      void* out16lo, spe_stream_t stream);

/* ---- out[c] (+)= sum_r in[r*ld + c] (bias gradients; autograd of nn.Linear bias; sums of split-K slabs).
 * accumulate = 1: added to what out holds (a zeroed buffer or a running sum); 0: out is overwritten (a weight gradient written
 * into its all-reduce bucket view needs no zeroing of the bucket beforehand). */
int spe_colsum(const float* in, float* out, long R, int C, long ld, int accumulate, spe_stream_t stream);

/* ---- LayerScale residual out = x + s_b*gamma*y (reference models/cait.py:413-416; s_b = the
 * per-sample DropPath keep scale or null).  bwd: dy = s_b*gamma*dout, dgamma += sum s_b*dout*y. */
int spe_layerscale_residual_fwd(const float* x, const float* y, const float* gamma, const float* sample_scale,
                                float* out, long R, int C, long rows_per_sample, spe_stream_t stream);
int spe_layerscale_residual_bwd(const float* dout, const float* y, const float* gamma, const float* sample_scale,
                                float* dy, float* dgamma, long R, int C, long rows_per_sample, spe_stream_t stream);
/* spe_layerscale_residual_bwd16: the same backward when the branch ends in a Linear on the bf16-copy GEMMs (proj / fc2 of
 * the backbone block, cait.py:390,412): dy = gamma * dout is emitted only as the bf16 operands of that Linear's backward
 * GEMMs - dy16 [R][C] and dy16T [C][ldt] (ldt = R rounded up to 64, padding zero) - with db[c] += sum_r dy (the Linear's
 * bias gradient, fp32 before rounding) and dgamma[c] += sum_r dout * y.  No per-sample scale (drop_path = 0).  y_f16 != 0: y holds
 * IEEE fp16 [R][C] (written by spe_gemm_bf16nt_ex with half_flags bit 0: y only ever enters this sum). */
int spe_layerscale_residual_bwd16(const float* dout, const void* y, int y_f16, const float* gamma, void* dy16, void* dy16T, long ldt,
                                  float* db, float* dgamma, long R, int C, spe_stream_t stream);
/* ... when the forward was out = x + s_b * gamma * dropout(y) (spe_gemm_bf16nt_exd): dy = s_b * keep * gamma * dout, dgamma += sum s_b * keep * dout * y. */
int spe_layerscale_residual_bwd16d(const float* dout, const void* y, int y_f16, const float* gamma, void* dy16, void* dy16T, long ldt,
                                   float* db, float* dgamma, long R, int C, float p_drop, uint64_t seed, uint64_t offset,
                                   const float* sample_scale, long rows_per_sample, spe_stream_t stream);

/* ---- activation backward: mode 1 ReLU (aux = forward output), mode 2 GELU (aux = pre-activation);
 * autograd of F.relu (transformer.py:32,287,424) and nn.GELU (timm Mlp). n % 4 == 0. */
int spe_act_bwd(const float* dy, const float* aux, float* dx, long n, int mode, spe_stream_t stream);

/* ---- norm(x + dropout(z)): the post-norm residual sites of the DETR encoder / decoder layers (reference
 * models/transformer.py:279-287, 384-386, 420-421, 426-427) in one pass each way.  fwd: sum = x + z*keepscale(row*C+c) (kept:
 * it is LayerNorm's input), y / mean / rstd as spe_layernorm_fwd.  bwd: ds = LayerNorm backward (gradient of x and of the sum),
 * dz = ds*keepscale with the same mask (not written when p == 0: the branch gradient is ds); dgamma / dbeta pre-zeroed; dy2 (optional): the
 * gradient of a second consumer of y - a post-norm layer's output feeds a Linear AND the next residual - added to dy while the row is loaded. */
int spe_layernorm_res_fwd(const float* x, const float* z, const float* gamma, const float* beta, float* sum, float* y,
                          float* mean, float* rstd, long R, int C, float eps, float p, uint64_t seed, uint64_t offset,
                          spe_stream_t stream);
int spe_layernorm_res_bwd(const float* dy, const float* dy2, const float* sum, const float* gamma, const float* mean, const float* rstd,
                          float* ds, float* dz, float* dgamma, float* dbeta, long R, int C, float p, uint64_t seed,
                          uint64_t offset, spe_stream_t stream);

/* ---- dropout y = x*keepscale (nn.Dropout at cait.py:387,391, transformer.py:266-288, timm Mlp);
 * the backward is the same call on dy. */
int spe_dropout(const float* x, float* y, long n, float p, uint64_t seed, uint64_t offset, spe_stream_t stream);

/* ---- patch gather for the 16x16 stride-16 patch-embed conv (reference models/cait.py:518-528):
 * img[B,Cin,Hi,Wi] -> cols[B*(Hi/P)*(Wi/P), Cin*P*P] in conv-weight column order. */
int spe_patchify(const float* img, float* cols, int B, int Cin, int Hi, int Wi, int P, spe_stream_t stream);

/* ---- D[b][h][q] = sum_d x[b][q][h][d] y[b][q][h][d] for contiguous fp32 [B, L, H, dh] tensors (16-B aligned when dh % 4 == 0): the row term
 * rowsum(dO . O) of the softmax backward of the decoder's cross attention (autograd of reference models/attention.py:277-383). */
int spe_rowdot(const float* x, const float* y, float* D, int B, int L, int H, int dh, spe_stream_t stream);

/* ---- out = a + b[(i mod period)] (adds the interpolated pos-embed table, cait.py:623-624). */
int spe_add_rows(const float* a, const float* b, float* out, long n, long period, spe_stream_t stream);

/* ---- bicubic (A=-0.75, align_corners=false) resize of the learned position grid, token-major:
 * forward: src[gh*gw][C] -> dst[h*w][C]; backward=1: src = d(out)[h*w][C], dst = d(in)[gh*gw][C] PRE-ZEROED.
 * Reference models/cait.py:598-613 (F.interpolate(mode='bicubic')). */
int spe_bicubic(const float* src, float* dst, int gh, int gw, int h, int w, int C, int backward, spe_stream_t stream);

/* ---- Hungarian matcher cost (reference models/matcher.py:62-83 + util/box_ops.py:33-74):
 * logits[L,B,Q,Kc], boxes[L,B,Q,4] (cxcywh), targets concatenated over images with prefix
 * offsets toff[B+1].  cost[l] = packed concat over b of row-major [Q, M_b] blocks:
 * w_bbox*L1 + w_class*(pos_focal-neg_focal)[label] - w_giou*GIoU.  *err is OR-ed with 1 if a
 * degenerate box is seen (the reference's host assert, box_ops.py:64-65). */
int spe_matcher_cost(const float* logits, const float* boxes, const int* tgt_ids, const float* tgt_boxes,
                     const int* toff, int total_targets, float* cost, int* err, int L, int B, int Q, int Kc,
                     float w_class, float w_bbox, float w_giou, spe_stream_t stream);

/* ---- the assignment itself, on the device (reference models/matcher.py:83-86: scipy.optimize.linear_sum_assignment
 * per image on the host).  cost / toff as produced for spe_matcher_cost; every image needs M_b <= Q <= 1024 (-2
 * otherwise: fall back to the host).  Writes, for problem (l, b), M_b triples at offset l*total + toff[b] in ascending
 * query order: srow = (l*B+b)*Q + q, gidx = toff[b] + j (int64 each), lidx = l (int32).  fp64 potentials like SciPy.
 * err (device int, may be null; the flag word of spe_matcher_cost): bit 1 (value 2) is set when a problem has a row whose
 * remaining costs are all NaN / infinite (SciPy raises ValueError there); that problem gets the identity assignment. */
int spe_hungarian(const float* cost, const int* toff, long* srow, long* gidx, int* lidx, int* err, int L, int B, int Q,
                  spe_stream_t stream);
/* One-to-many target jitter (reference models/conditional_detr.py:409-431): out [M][ratio][4] - for every cxcywh box up to ratio - 1 of
 * its candidates scale[m][c][:] * box[m][:] (c < ncand, attempt order) whose IoU with the box exceeds 0.7, then the box itself for the
 * picks that found none and for the last row.  scale [M][ncand][4]: the uniform draws (the caller's generator), 16-B aligned. */
int spe_jitter_pick(const float* box, const float* scale, float* out, int M, int ncand, int ratio, spe_stream_t stream);

/* ---- weighted sigmoid focal loss (reference models/conditional_detr.py:468-494, 504-535):
 * logits[L*rows_per_l, Kc]; tclass[row] in [0,Kc] (Kc = no object); roww[row] row weight or
 * null.  loss[l] += sum (pre-zeroed by caller); grad = d(sum)/d(logit); argmax = top-1 class. */
int spe_focal_loss(const float* logits, const int* tclass, const float* roww, float* grad, float* loss,
                   int* argmax, int L, long rows_per_l, int Kc, float alpha, float gamma, spe_stream_t stream);

/* ---- matched-pair box losses (reference models/conditional_detr.py:300-319, 537-560):
 * pair i = (row srow[i] of pred_boxes[*,4], tbox[i], weight w[i] or null, layer lidx[i]).
 * sums[l][0] += w*L1, sums[l][1] += w*(1-GIoU) for l < L (pairs added in index order); g_l1/g_giou = d/d(pred cxcywh).
 * bwd scatter-adds c1[l]*g_l1 + c2[l]*g_giou into dpred rows. */
int spe_box_loss(const float* pred_boxes, const long* srow, const float* tbox, const float* w, const int* lidx,
                 float* sums, float* g_l1, float* g_giou, long n, int L, spe_stream_t stream);
int spe_box_loss_bwd(const long* srow, const int* lidx, const float* g_l1, const float* g_giou, const float* c1,
                     const float* c2, float* dpred, long n, spe_stream_t stream);

/* ---- memory side of the decoder's conditional cross attention for ALL layers (reference models/transformer.py:389-419):
 * spe_kv_frags turns the fp16 outputs of the two stacked projection GEMMs (spe_gemm_bf16nt with act bits 8 + 9) -
 *   ym16 [B*S][ldm]: column block 2l = ca_kcontent_proj_l(memory), 2l + 1 = ca_v_proj_l(memory);  yp16 [B*S][ldp]: block l =
 *   ca_kpos_proj_l(pos); d = H * dh columns per block -
 * into the operand fragments spe_mha_fwd / spe_mha_bwd consume, stacked over the layers ([L][B][H][ceil(S/16)] records):
 *   Kf  fp16, 32-wide steps of the 2 dh key dims [k_content | k_pos] of a head (spe_attn_pack_multi kind 2 + 16 layout)
 *   V16 fp16, 16-wide (kind 1 + 16);   K16 bf16, 16-wide (kind 1) and Vf bf16, 32-wide steps (kind 2): backward only, both or neither.
 * No fp32 key / value tensor, concatenation or per-layer pack.  dh % 8 == 0, dh <= 64, ldm / ldp % 8 == 0.
 * spe_kv_grad_scatter: dk [B,S,H,2 dh] / dv [B,S,H,dh] (fp32, spe_mha_bwd) of one layer -> bf16 column blocks 2l (content half of
 *   dk), 2l + 1 (dv) of dYm [B*S][ldm] and block l (pos half of dk) of dYp [B*S][ldp]: the stacked dY operands of the projection
 *   GEMMs' backward. */
int spe_kv_frags(const void* ym16, long ldm, const void* yp16, long ldp, void* Kf, void* V16, void* K16, void* Vf,
                 int L, int B, int S, int H, int dh, spe_stream_t stream);
int spe_kv_grad_scatter(const float* dk, const float* dv, void* dYm, long ldm, void* dYp, long ldp, int layer, int B, int S,
                        int H, int dh, spe_stream_t stream);
/* column sums of a bf16 matrix [R][ld] made of nblk (<= 32) column blocks of blkC columns, block i into its own fp32 vector outs[i]
 * (HOST array of device pointers): the bias gradients of the stacked projections, fixed summation order. */
int spe_colsum_bf16_blocks(const void* x, long ld, long R, int nblk, int blkC, float* const* outs, int accumulate, spe_stream_t stream);

/* ---- flash-style multi-head attention (reference models/attention.py:277-383; nn.MultiheadAttention core of the
 * encoder, models/transformer.py:275-277): softmax(scale q k^T + key_padding_mask), dropout, . v, forward and backward
 * without the [B,H,Lq,Lk] score tensor.  Operands are fragments from spe_attn_pack_multi: kind 2 (32-wide steps) of
 * q*scale*log2(e), k, v, dO -> Qf, Kf, Vf, dOf ; kind 1 (16-wide) of v, k, q*scale*log2(e), dO -> V16, K16, Q16, dO16.
 * Element formats: Qf, Kf and V16 FP16 (kinds + 16: the forward operands), Vf, dOf, K16, Q16, dO16 BF16.
 * q/k head dim <= 96, v head dim <= 64; mask [B,Lk] uint8 (1 = padded) or NULL; Philox dropout on element index
 * ((b*H+h)*Lq + q)*ld4 + key, ld4 = Lk rounded up to 4 (the stream spe_softmax_fwd draws from); the forward records the
 * keep flags in keepbits (B*H*ntq*ntk*4 64-bit words, needed when p_drop > 0) and the backward reads them.
 * spe_mha_plan: number of key chunks the forward / dQ kernels split the keys into (few query tiles -> many chunks).
 * spe_mha_fwd: Opart [B*H*ntq*nch][ceil(dv/16)][64][4] and ML [B*H*ntq*nch][16][2] floats of workspace ->
 *   O [B,Lq,H*dv], LSE [B,H,Lq] (log2 domain).  spe_mha_bwd: D [B,H,Lq] = rowsum(dO.O); dq [B,Lq,H,dk] must be
 *   zero-initialised when nch > 1 and dq_ws == NULL (atomic accumulation over chunks); with dq_ws [nch][B*Lq*H*dk] every
 *   chunk writes its own partial slab instead (sum them with spe_colsum: no atomics, fixed order) and dq is not touched;
 *   dk [B,Lk,H,dk], dv [B,Lk,H,dv] are overwritten. */
int spe_mha_plan(int B, int H, int Lq, int Lk, int* nch);
int spe_mha_fwd(const void* Qf, const void* Kf, const void* V16, const void* mask, float* Opart, float* ML, float* O,
                float* LSE, void* keepbits, int B, int H, int Lq, int Lk, int dk, int dv, int nch, float p_drop,
                uint64_t seed, uint64_t offset, spe_stream_t stream);
int spe_mha_bwd(const void* Qf, const void* Kf, const void* Vf, const void* dOf, const void* K16, const void* Q16,
                const void* dO16, const void* mask, const float* LSE, const float* D, const void* keepbits, float* dq,
                float* dq_ws, float* dk, float* dv, int B, int H, int Lq, int Lk, int dk_dim, int dv_dim, int nch, float scale,
                float p_drop, spe_stream_t stream);

/* ---- sine position embedding of the padded feature map (reference models/position_encoding.py:37-57):
 * mask [B,h,w] uint8 (1 = padded), dim_t[npf] = temperature^(2*(k/2)/npf), out [B,h,w,2*npf] fp32 (row features first). */
int spe_pos_sine(const void* mask_u8, const float* dim_t, float* out, int B, int h, int w, int npf, float scale,
                 float eps, int normalize, spe_stream_t stream);

/* ---- per-class greedy NMS (reference engine_loc.py:154-174: torchvision.ops.nms(boxes, scores, 0.5) per predicted
 * class, results concatenated in ascending class order).  boxes [nimg, nmax, 4] xyxy and labels [nimg, nmax] (int64)
 * ALREADY ordered by (label ascending, score descending) per image; counts[img] valid detections (NULL: nmax).
 * keep[img][i] = 1 iff detection i survives (suppress IoU > iou_threshold within a class).  nmax <= 4096. */
int spe_nms_sorted(const float* boxes, const long* labels, const int* counts, unsigned char* keep, int nimg, int nmax,
                   float iou_threshold, spe_stream_t stream);

/* ---- CAM -> pseudo boxes (reference cams_deit.py:9-13 resize_cam, :61-96 get_multi_bboxes; engine.py:356-398).
 * spe_cam_prepare (device): M class maps cams[M][h][w] -> thresholded uint8 images out[M][rows][cols]: bilinear resize
 * (cv2.INTER_LINEAR convention), min-max normalisation, (x*255) truncated to uint8, THRESH_TOZERO at
 * int(cam_thr * max).  minmax: 2*M floats of workspace.  Synchronises the stream once (workspace initialisation).
 * spe_cam_contour_boxes (HOST function, host pointers): borders of the non-zero pixels of one image (Suzuki-Abe border
 * following, 8-connected, outer and hole borders = cv2.findContours RETR_TREE), polygon areas (cv2.contourArea) and the
 * boxes [x, y, x+w, y+h] (cv2.boundingRect) of every border with area >= area_ratio * largest, largest first;
 * [0,0,1,1] when there is none.  -5: more than max_boxes boxes. */
int spe_cam_prepare(const float* cams, int M, int h, int w, int rows, int cols, float cam_thr, float* minmax,
                    void* out, spe_stream_t stream);
int spe_cam_contour_boxes(const void* img, int rows, int cols, float area_ratio, int* boxes, int max_boxes, int* nboxes);

/* ---- optimiser step on flat buffers (reference engine.py:161-165: clip_grad_norm_(params, 0.1) + AdamW.step(),
 * parameter groups of main.py:177-191).  spe_sqnorm_partials: partials[b] = sum g^2 over the b-th of nblocks chunks.
 * spe_adamw_flat: clip = min(1, max_norm / (sqrt(sum partials) + 1e-6)) (max_norm <= 0: no clipping), g *= clip,
 * then torch.optim.AdamW's update with bias corrections bias_c1 = 1 - beta1^t, bias_c2 = 1 - beta2^t; element i uses
 * (lr, weight decay) of the segment s with seg_end[s-1] <= i < seg_end[s] (nseg <= 64).  write_grad != 0 stores the
 * clipped gradient back (what clip_grad_norm_ leaves in .grad).  Buffers 16-B aligned.  grad_scale: g is first multiplied
 * by it (1/world after a SUM all-reduce: DistributedDataParallel's gradient averaging, reference main.py:172, folded into
 * this launch); the clip norm is that of the scaled gradient. */
int spe_sqnorm_partials(const float* g, long n, float* partials, int nblocks, spe_stream_t stream);
int spe_adamw_flat(float* p, float* g, float* m, float* v, long n, const long* seg_end, const float* seg_lr,
                   const float* seg_wd, int nseg, float beta1, float beta2, float eps, float bias_c1, float bias_c2,
                   const float* partials, int npartials, float max_norm, int write_grad, float grad_scale,
                   spe_stream_t stream);

/* ---- diagnostic (never launched by the product): keep nwg workgroups resident for `micros` microseconds, streaming copies
 * through buf (buf_floats floats, 16-B aligned; NULL: idle spinning) - the one-GPU proxy for the CU / HBM share of an RCCL ring
 * running beside the backward (reference main.py:172: DistributedDataParallel overlaps its all-reduce with the backward);
 * tools/dp_proxy.py measures the slowdown of the step under it. */
int spe_occupy(int nwg, long micros, float* buf, long buf_floats, spe_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPE_HIP_H */
