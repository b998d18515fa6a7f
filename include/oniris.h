/* liboniris_hip.so -- C ABI of the MI355X-native Oniris denoiser-step kernels (gfx950 only).
 *
 * The reference (Francesco215/autoregressive_diffusion) has no FFI layer: its hot path is PyTorch dispatch from
 * edm2/networks_edm2.py, edm2/conv.py and edm2/attention/.  Each entry point below names the reference call
 * site(s) it replaces (file:line relative to the reference root).  Conventions:
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless marked [host];
 *   - the caller allocates every buffer; no hidden allocation, no stream synchronisation; process-wide mutable state is
 *     limited to three host-side knobs, each named where it is declared: oniris_set_cu_reserve (CUs left to RCCL),
 *     oniris_set_ew_nt_bytes (non-temporal threshold) and the diagnostic dispatch census (oniris_census);
 *   - returns 0 (ONIRIS_OK) or a negative code; oniris_last_error() gives a thread-local message;
 *   - activations are channels-last bf16: a frame-slot tensor is [N][H][W][C] (C % 16 == 0 except outputs);
 *     frame-slot index n = b*(S*T) + s*T + t  (S = 2 clean|noised in training -- the reference's '(b s t)'
 *     order, edm2/conv.py:79 -- S = 1 in eval);
 *   - `stream` is a hipStream_t passed as void*.
 */
#ifndef ONIRIS_H
#define ONIRIS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* oniris_stream_t;

const char* oniris_last_error(void);
int oniris_abi_version(void);   /* 14.  13 -> 14: oniris_set_ew_nt_bytes, oniris_census / oniris_census_read (diagnostics), the fp32 verification path (oniris_conv_f32 / wgrad_f32 / attn_f32_*); no signature changed; 12 -> 13: oniris_dart_input(+ cpad: the packed input is 32 channels wide in the product, so that the stem conv runs on the
                                 * streaming kernels of the 32-channel level); 11 -> 12: oniris_set_cu_reserve; 10 -> 11: OnirisConvArgs.ctx_prod / ctx_prod_mode (appended fields); 9 -> 10: OnirisConvArgs.clip_flag,
                                 * oniris_gconv_bwd_fused(+ clip_flag, coef_own_scaled), oniris_qkv_norm_hd / _hd_bwd / oniris_rope_hd       */
/* Measurement aid: arm a pair of HIP events (hipEvent_t created with timing); the next MFMA conv / weight-gradient /
 * scheduled attention-forward kernel this THREAD launches records its own begin and end into them (hipExtLaunchKernel:
 * the dispatch's timestamps, as rocprofv3 reports them; events recorded around a launch also time the kernel boundary).
 * oniris_profile_disarm returns 1 when the pair was not consumed (the entry point took a path without the hook).       */
int oniris_profile_arm(void* start_event, void* stop_event);
int oniris_profile_disarm(void);
/* CUs the persistent kernels (one workgroup per CU: the LDS-DMA convolutions) leave free from now on: the data-parallel
 * wrapper sets k while a gradient exchange is in flight beside the backward kernels, so that RCCL's workgroups find CUs of
 * their own instead of delaying a persistent workgroup by a whole tile run (the reference: torch DDP's bucketed all-reduce
 * overlapping backward, cs_train.py:53-54,108-114).  Process-wide, host-side, takes effect at the next launch; k is rounded
 * up to a multiple of 8 (one CU per XCD).  Returns the previous value, or a negative error code.                        */
int oniris_set_cu_reserve(int k);
/* Size in bytes from which the single-pass kernels and the conv output stores stream a tensor with non-temporal accesses
 * (default 96 MiB, or ONIRIS_EW_NT_MB at first use; bytes < 0: never).  Process-wide, host-side, takes effect at the next
 * launch; returns the previous threshold.  The results do not depend on it (same arithmetic, other cache policy): the test
 * suite sets 0 to put every non-temporal instantiation under the oracle on oracle-sized tensors.                       */
long long oniris_set_ew_nt_bytes(long long bytes);
/* Dispatch census (diagnostic).  oniris_census(1) clears the list and starts noting every kernel launch of this library;
 * oniris_census(0) stops.  oniris_census_read writes one line per distinct launch kind, "<launches>\t<kernel instantiation>
 * [ [tag]]\n" (demangled name with its template arguments; the tag marks variants picked at run time inside one
 * instantiation, e.g. "nt-stores"), NUL-terminated, truncated to cap; returns the bytes needed (call with NULL, 0 first).
 * tests/test_zz_dispatch_coverage.py: the set the bench's timed regions launch must be a subset of the set the
 * oracle-comparing tests launched.                                                                                        */
int oniris_census(int on);
long long oniris_census_read(char* buf /* [host] */, long long cap);
/* sizeof(OnirisWeightDesc, OnirisConvArgs, OnirisWgradArgs, OnirisAttnArgs) for binding self-checks */
int oniris_struct_sizes(int32_t* out4 /* [host] */);

/* ---------------------------------------------------------------------------------------------------------------
 * Mask tables [host, int32] -- bit-exact replacement of make_train_mask / make_infer_mask
 * (edm2/attention/attention_masking.py:27-53, 64-90).
 * oniris_train_mask: writes kv_num_blocks[2*nb] and kv_indices[2*nb][2*nb] for ONE (batch, head) (the reference
 * repeats the same table over b,h).  Returns the number of row blocks 2*nb (>0), 0 when the reference returns
 * None (T*P % 128 != 0 with P < 128), <0 on error.  Pass NULL outputs to query the size.  *block_size receives
 * the BlockMask BLOCK_SIZE (P if P >= 128 else 128).
 * oniris_infer_mask: same for the causal prefill mask; returns nb, 0 for the 'score_mod'/'dense' fall-backs
 * (pure frame-causal mask_mod, no table).
 * oniris_mask_transpose: inverts a kv table into the q table used by the dK/dV kernel (for each kv block: the
 * q blocks that list it, ascending).
 */
int oniris_train_mask(int n_frames, int image_size, int32_t* kv_num_blocks, int32_t* kv_indices, int* block_size);
int oniris_infer_mask(int n_frames, int image_size, int32_t* kv_num_blocks, int32_t* kv_indices, int* block_size);
int oniris_mask_transpose(int n_rows, int n_cols, const int32_t* kv_num_blocks, const int32_t* kv_indices,
                          int32_t* q_num_blocks, int32_t* q_indices);

/* ---------------------------------------------------------------------------------------------------------------
 * Weights: forced normalisation + bf16 packing of ALL weights in one launch, and the backward through the
 * normalisation.  Replaces NormalizedWeight.forward (edm2/conv.py:14-21) and the weight casts in
 * MPConv.forward / MPCausal3DGatedConv.forward (edm2/conv.py:37,63).
 * Descriptor table (device array, built once by the host):                                                     */
typedef struct OnirisWeightDesc {
  float* w;        /* fp32 parameter (cout, cin, taps) row-major == the reference (O,I[,kt],kh,kw) tensor        */
  float* grad;     /* fp32 gradient, same shape; oniris_weight_bwd ACCUMULATES into it                           */
  void* wf;        /* bf16 packed forward weight  [taps][CoutP][CinP]   (ci contiguous)                          */
  void* wb;        /* bf16 packed dgrad weight    [taps][CoutPb][CinPb] = flipped/transposed copy (may be NULL)  */
  void* dwp;       /* bf16 split-K slabs [nsplit_cap][CoutP][taps][CinP]: oniris_conv_wgrad workgroup column s   *
                    * overwrites slab s with plain stores (its fp32 partial sum rounded once); oniris_weight_bwd  *
                    * adds the first *nsplit slabs in fp32                                                         */
  float* dws;      /* fp32 [CoutP][taps][CinP]: the sum of the slabs (scratch of oniris_weight_bwd)               */
  int32_t cout, cin, taps, kt;     /* taps = kt*kh*kw (1, 9 or 18); kt = temporal taps (1 or 2)                  */
  int32_t CoutP, CinP;             /* CoutP = roundup(cout,32), CinP = roundup(cin,64)                           */
  int32_t CoutPb, CinPb;           /* CoutPb = roundup(cin,32), CinPb = roundup(cout,64)                         */
  int32_t row_start;               /* prefix sum of cout over the table                                          */
  int32_t perm3;                   /* 1: attn_qkv rows (m c s) are packed as (s m c) (attention_modules.py:48)   */
  float gain;                      /* static gain folded into the packed weight                                  */
  int32_t nsplit_cap;              /* slabs allocated behind dwp                                                 */
  int32_t* nsplit;                 /* device int: slabs written since the last oniris_weight_prep (which resets  *
                                    * it to 0); set by oniris_conv_wgrad, read by oniris_weight_bwd              */
  int32_t tile_start;              /* prefix sum of ceil(cout/32) over the table (oniris_weight_prep: one         *
                                    * workgroup per 32 packed rows)                                               */
  int32_t pad_;
} OnirisWeightDesc;

/* total_rows = sum of cout, total_tiles = sum of ceil(cout/32) over the table (two kernels: normalise + forward
 * packing per row, then the transposed dgrad packing per 32-row tile)                                             */
int oniris_weight_prep(const OnirisWeightDesc* descs, int ndesc, int total_rows, int total_tiles, int training,
                       oniris_stream_t stream);
int oniris_weight_bwd(const OnirisWeightDesc* descs, int ndesc, int total_rows, oniris_stream_t stream);

/* DART training input and loss (edm2/loss.py:17-47 with Precond.forward, networks_edm2.py:278-297) as three passes
 * instead of ~25 activation-sized fp32 elementwise launches.  images [B][T][C][H][W], noise [B][S*T][C][H][W], sigma
 * [B][S*T] fp32; slot n = (b, s, t), s = 0 clean | 1 noised (S = 1 in 2-D steps); x[n] = images[b,t] + sigma*noise is
 * never materialised.
 *   dart_input:    xcl bf16 [B*S*T][H][W][cpad] = c_in * x, channel C = 1 (ones channel), the rest 0        (C < 16)
 *   dart_loss:     losses[b][t] = mean_{c,h,w} (c_skip*x + c_out*out_gain*F - images)^2 of the noised half; F = the
 *                  UNet's channels-last output bf16 [B*S*T][H][W][8], out_gain a device scalar             (C <= 8)
 *   dart_loss_bwd: dF (bf16, zero for clean slots) and dgain_part[b][t] (sum them for d out_gain) from dlosses[b][t] */
int oniris_dart_input(const float* images, const float* noise, const float* sigma, void* xcl, int B, int S, int T, int C,
                      int H, int W, float sigma_data, float* c_noise_out, int cpad /* ABI 13: channels of xcl (16 ... 64, % 8) */,
                      oniris_stream_t stream);
/*   c_noise_out (may be NULL): [B*S*T] fp32 = log(sigma) / 4, the UNet's noise conditioning (networks_edm2.py:291), written by
 *   the same launch (the sampler's evaluations save two tiny launches each)                                           */
int oniris_dart_loss(const void* F, const float* images, const float* noise, const float* sigma, const float* out_gain,
                     float* losses, int B, int S, int T, int C, int H, int W, float sigma_data, oniris_stream_t stream);
int oniris_dart_loss_bwd(const void* F, const float* images, const float* noise, const float* sigma,
                         const float* out_gain, const float* dlosses, void* dF, float* dgain_part, int B, int S, int T,
                         int C, int H, int W, float sigma_data, oniris_stream_t stream);

/* Tail of EDM2Loss.__call__ + MultiNoiseLoss.add_data without a host round trip (replaces reference edm2/loss.py:32-46
 * and edm2/loss_weight.py:30-39,104-111,126-131):  per (b, t) of the noised half, sigma read at sigma[b*sig_pitch + sig_off + t]:
 *   l = mse[b][t] * (sigma^2 + sd^2) / (sigma*sd)^2;   m = 10^(Fourier series of log10 sigma, coef [2*nterms-1]);
 *   out[0] = mean(l / m) (the training loss), out[1] = mean(l) (the un-weighted loss), dcoef[b][t] = d out[0] / d mse[b][t];
 *   (sigma, l, t) is appended to the history rings ring_sigma / ring_loss / ring_pos [cap] behind the device-side entry
 *   counter *count (entry e lives in slot e % cap); ring_sigma == NULL: no logging (ranks other than 0, loss_weight.py:33). */
int oniris_loss_tail(const float* mse, const float* sigma, const float* coef, float* out, float* dcoef, float* ring_sigma,
                     float* ring_loss, int* ring_pos, long long* count, int cap, int B, int T, int sig_pitch, int sig_off,
                     int nterms, float sigma_data, oniris_stream_t stream);

/* Eval-side counterparts (edm2/sampler.py: 31 evaluations per generated frame, every launch counts):
 * oniris_dart_input with noise == NULL packs c_in * x (Precond.forward's input side, networks_edm2.py:287-291);
 * oniris_precond_out: D [N][C][H][W] fp32 = c_skip * x + c_out * out_gain * F  (F = raw channels-last UNet output
 *   [N][H][W][8] bf16, x fp32 like D, sigma [N]; networks_edm2.py:293-297);
 * oniris_gates: every Gating module of a net in one launch (edm2/conv.py:113-127): params [L][6] = mult0, mult1, off0,
 *   off1, min_gating, max_gating; nctx [L] frame counters (NULL = 0); c_noise [N]; position of slot n = n % T;
 *   outputs the gate coefficients ca, cb [L][N] of mp_sum(y2, y3, g) (utils.py:118-123).                              */
int oniris_precond_out(const void* F, const float* x, const float* sigma, const float* out_gain, float* D, int N, int C,
                       int H, int W, float sigma_data, oniris_stream_t stream);
/* oniris_sampler_update: the Euler / Heun update between two UNet evaluations of edm_sampler_with_mse (reference
 *   edm2/sampler.py:66-76), one fp32 launch over n elements:
 *   mode 0: d = (x_hat - x_pred) / t_a; x_out = x_hat + dt * d; d_io <- d
 *   mode 1: d' = (x_aux - x_pred) / t_a; x_hat <- x_out <- x_hat + dt * (0.5 * d_io + 0.5 * d')
 *   sigma_buf (nsig floats, may be NULL): filled with sigma_next (the sigma input of the evaluation that follows).      */
int oniris_sampler_update(int mode, float* x_hat, const float* x_pred, float* d_io, const float* x_aux, float* x_out, size_t n,
                          float t_a, float dt, float* sigma_buf, int nsig, float sigma_next, oniris_stream_t stream);
 /* oniris_embed_eval: the UNet's noise / label embedding (networks_edm2.py:204-216) in one fp32 launch: emb [N][cemb] bf16
 *   = mp_silu(mp_sum(MPConv_noise(MPFourier(c_noise)), MPConv_label(onehot(labels) * sqrt(L)), 1/3)); w_noise [cemb][cnoise],
 *   w_label [cemb][L] are the RAW fp32 parameters (normalised per row inside); labels / w_label NULL: no label term.      */
int oniris_embed_eval(const float* c_noise, const int64_t* labels, const float* freqs, const float* phases,
                      const float* w_noise, const float* w_label, void* emb, int N, int cnoise, int cemb, int label_dim,
                      oniris_stream_t stream);
int oniris_gates(const float* c_noise, const float* params, const int32_t* nctx, float* ca, float* cb, int L, int N, int T,
                 oniris_stream_t stream);

/* Training-side conditioning prelude with explicit adjoints (a few fused launches instead of ~280 torch elementwise
 * launches on kilobyte-sized tensors per step):
 * oniris_gates_bwd: adjoint of oniris_gates, dparams [L][6] from dca, dcb [L][N] (Gating, edm2/conv.py:113-127);
 * oniris_emb_scale: c [N][Ctot] fp32 = 1 + c_all * gain[seg[j]] for every Block at once (networks_edm2.py:78; c_all
 *   [N][Ctot] bf16 = the row-concatenated emb_linear GEMM, seg [Ctot] = Block of column j, gain [K] = the emb_gain's);
 *   oniris_emb_scale_bwd: dc_all bf16 and dgain_part [K][ONIRIS_EMB_BWD_CHUNKS] from dc (start [K+1] = first column of
 *   each Block; d gain[k] = the sum of row k of dgain_part, added by the caller: deterministic, no atomics);
 * oniris_embed_pre: the inputs of the embedding linears, four [N][cnoiseP] bf16 = MPFourier(c_noise) (utils.py:139-150),
 *   onehot [N][labelP] bf16 = one_hot(labels) * sqrt(L) (networks_edm2.py:209; NULL = no labels), zero-padded columns;
 * oniris_embed_post: emb = mp_silu(mp_sum(e1, e2, t)) (networks_edm2.py:210-212; e2 NULL: mp_silu(e1)), bf16 [n];
 *   oniris_embed_post_bwd: its adjoint (de1, de2 from demb).                                                          */
int oniris_gates_bwd(const float* c_noise, const float* params, const int32_t* nctx, const float* dca, const float* dcb,
                     float* dparams, int L, int N, int T, oniris_stream_t stream);
int oniris_emb_scale(const void* c_all, const float* gain, const int32_t* seg, float* c, int N, int Ctot,
                     oniris_stream_t stream);
#define ONIRIS_EMB_BWD_CHUNKS 16
int oniris_emb_scale_bwd(const float* dc, const void* c_all, const float* gain, const int32_t* start, void* dc_all,
                         float* dgain_part, int N, int Ctot, int K, oniris_stream_t stream);
int oniris_embed_pre(const float* c_noise, const int64_t* labels, const float* freqs, const float* phases, void* four,
                     void* onehot, int N, int cnoise, int cnoiseP, int label_dim, int labelP, oniris_stream_t stream);
int oniris_embed_post(const void* e1, const void* e2, void* emb, size_t n, float t, oniris_stream_t stream);
int oniris_embed_post_bwd(const void* demb, const void* e1, const void* e2, void* de1, void* de2, size_t n, float t,
                          oniris_stream_t stream);

/* Fused AdamW over flat fp32 buffers (the optimizer step of gym_train.py:105-106 / cs_train.py:117-118).       */
int oniris_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, float grad_scale, oniris_stream_t stream);
/* The whole optimizer side of a training step in one pass (gym_train.py:105-108): gradient-norm clipping
 * (torch.nn.utils.clip_grad_norm_: gradients scaled by min(1, max_norm / (||grad_scale*g|| + 1e-6)); gnorm_sq = device
 * scalar holding sum(g^2) from oniris_sqnorm, NULL = no clipping), AdamW, and the power-function EMA update of up to
 * two tracked copies (edm2/phema.py:101-106: ema += ema_w * (p_new - ema), ema_w = 1 - beta; NULL = not tracked).
 * step = the Adam step count of THESE parameters (bias correction); step == 0: the parameters received no gradient
 * (torch.optim skips a parameter whose .grad is None): p, m, v stay untouched, only the EMA copies follow.             */
int oniris_adamw_clip_ema(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int step, float grad_scale, const float* gnorm_sq,
                          float max_norm, float* ema0, float ema_w0, float* ema1, float ema_w1, oniris_stream_t stream);
/* out[0] <- sum of squares of g[0..n) ; out must hold 1 + ONIRIS_SQNORM_WS floats (out[1..] is workspace).
 * Deterministic (two-stage, no float atomics), never synchronises.                                               */
#define ONIRIS_SQNORM_WS 1024
int oniris_sqnorm(const float* g, size_t n, float* out, oniris_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Gated causal 3-D convolution as ONE implicit GEMM (MFMA bf16 -> fp32), forward and data-gradient.
 * Replaces F.conv2d (edm2/conv.py:41,74) + F.conv3d over two previous clean frames (edm2/conv.py:86) + the gated
 * mp_sum (edm2/conv.py:95) + the cat/stack/rearrange copies (edm2/conv.py:79-91); with taps == 1 it is the 1x1
 * conv / linear of MPConv.forward (edm2/conv.py:39-41).
 *   out[n] = coef_own[n] * conv(x[n], w_own) + coef_ctx[n] * sum_j conv(ctxframe(b, t + coff[j]), w_ctx[j])
 * ctxframe(b,f) = ctx[b*ctx_bstride + f] if 0 <= f < ctx_T else a frame filled with ctx_fill (zero outside the
 * image).  Forward (train): ctx = x, coff = {-2,-1}, ctx_fill = 1 (edm2/conv.py:68).  Data-gradient: x = dout,
 * ctx = dy3, coff = {+2,+1}, ctx_fill = 0, packed weights = desc.wb.
 * Epilogues (fused magnitude-preserving blocks, edm2/networks_edm2.py:73-93):
 *   ONIRIS_EPI_NONE      out = v
 *   ONIRIS_EPI_EMB_SILU  out = v ; out2 = silu(v * escale[n][co]) / 0.596        (escale = 1 + emb_gain*emb_linear(emb))
 *   ONIRIS_EPI_MPSUM     out = clip( ta * res[n][p][co] + tb * v , +-clip )     (clip <= 0: no clipping);
 *                        out2 (optional) = v, the raw conv output (kept for the gate gradient)
 */
enum { ONIRIS_EPI_NONE = 0, ONIRIS_EPI_EMB_SILU = 1, ONIRIS_EPI_MPSUM = 2 };

typedef struct OnirisConvArgs {
  const void* x;          /* bf16 [B*S*T][H][W][Cin]                                                              */
  const void* ctx;        /* bf16 [B*ctx_bstride][H][W][Cin] or NULL (no context path)                            */
  const void* w_own;      /* bf16 packed [taps][CoutP][CinP]                                                      */
  const void* w_ctx;      /* bf16 packed [2*taps][CoutP][CinP] or NULL                                            */
  void* out;              /* bf16 [B*S*T][H][W][Cout]                                                             */
  const float* coef_own;  /* [B*S*T] or NULL (= 1)                                                                */
  const float* coef_ctx;  /* [B*S*T] or NULL (= 1)                                                                */
  int32_t B, S, T, H, W;
  int32_t Cin, CinP, Cout, CoutP;
  int32_t taps;           /* 9 (3x3, zero padding 1) or 1                                                         */
  int32_t ctx_bstride, ctx_T, coff0, coff1;
  float ctx_fill;
  int32_t epi;
  const void* res;        /* bf16 [B*S*T][H][W][Cout]   (EPI_MPSUM)                                               */
  const void* escale;     /* fp32 [B*S*T][Cout] (row pitch: escale_pitch)  (EPI_EMB_SILU): per-(frame, channel) multiplier */
  const float* emb_gain;  /* reserved (unused)                                                                    */
  void* out2;             /* bf16 like out              (EPI_EMB_SILU: activation; EPI_MPSUM: optional raw output) */
  float ta, tb, clip;
  void* ctx_out;          /* optional bf16 [B*T][H][W][Cout]: the un-gated context product y3 (for d gate)        */
  int32_t big_tile;       /* variant: 0 = 4-wave register-staged kernels, 1/2 = 8-wave ones where they fill the chip /
                           * always, >= 3 = persistent LDS-DMA kernel (csrc/conv_glds.h) wherever the shape allows,   *
                           * >= 4 = + the streaming kernels of the 32-channel level (conv_stream.h, conv_plain_stream.h); *
                           * diagnostic bits: 16 = no conv_eval1_kernel, 32 = copy issue of conv_glds at the phase start, 64 = 64-channel output tiles
                           * in the few-tile 1x1 launches too (conv_fwd_s1.hip), 128 = no conv_plain_stream_kernel, 256 = conv_eval1_kernel
                           * always with 32 output channels per workgroup (csrc/conv_eval1.h), 512 = no conv1x1_few_kernel (the few-tile
                           * 1x1 launches on conv_fwd_kernel's 32-channel tiles)            */
  int32_t escale_pitch;   /* floats between consecutive rows of escale (0 = Cout): a UNet's emb-scales are column
                           * blocks of ONE [B*S*T][sum Cout] GEMM output, read in place                            */
  /* Optional split-K workspace (caller-allocated, reusable by consecutive launches on one stream): when given and
   * the launch has <= 64 tiles (a single generated frame in the sampler, edm2/sampler.py:12-85), the
   * K = taps*Cin*(1 or 3 phases) loop of a tile is dealt to several workgroups which write fp32 partial sums here;
   * a second launch adds them in slice order and runs the epilogue.                                               */
  float* splitk_ws;
  size_t splitk_ws_bytes;
  /* Optional (EPI_MPSUM with clip > 0, ABI 10): device int that the launch ORs 1 into when the clip changed at least one
   * element.  Zero it before the launch; kernels that do not support it leave it alone (the caller knows which ones do:
   * csrc/conv_glds.h, csrc/conv_stream.h).  The backward pre-pass reads it: activations of a magnitude-preserving net
   * practically never reach the clip, and then the gradient needs no mask -- no read of the clipped output, no masked copy
   * of the incoming gradient (oniris_gconv_bwd_fused, mode 2).                                                      */
  int32_t* clip_flag;
  /* Optional (ABI 11; the one-frame cached evaluation only: S == 1, T == 1, ctx = the cached pair, csrc/conv_eval1.h): the
   * un-gated context product  y3[n][p][co] = sum_j conv(ctxframe(b, coff[j]), w_ctx[j])  in fp32, [B][H][W][Cout].  The sampler
   * evaluates the net 31 times per generated frame against the SAME cached pair (edm2/sampler.py:50-76 inside one frame of
   * :36-85; the reference recomputes F.conv3d over the cached frames every time, edm2/conv.py:84-86): y3 does not change
   * between those evaluations, only the gate coefficient that scales it does.
   *   ctx_prod_mode 0: ctx_prod unused.
   *                 1: all phases as usual, and y3 is ALSO stored to ctx_prod.
   *                 2: y3 is READ from ctx_prod; the context phases (two thirds of the weight stream) are skipped.  Bit-identical
   *                    to mode 0 / 1 on the same inputs (the fp32 sum is stored before the gate touches it).
   *                 3: ONLY y3 is computed and stored; out / out2 / res / x are not touched (x and out may be NULL).
   * A launch with ctx_prod_mode != 0 that the one-frame kernel cannot serve fails with ONIRIS_EUNSUPPORTED.        */
  float* ctx_prod;
  int32_t ctx_prod_mode;
  /* Optional (ABI 12; 1x1 convs through the register-staged kernel only: evaluation-sized launches): the input is the
   * magnitude-preserving CONCATENATION of two tensors, formed on the way in -- channels [0, x_split) come from x ([pos][x_split])
   * times cat_w1, channels [x_split, Cin) from x2 ([pos][Cin - x_split]) times cat_w2, each product rounded to bf16 (utils.py:128-134:
   * what oniris_act_fwd stores as xo) -- and act_out [pos][Cin] receives mp_silu of that value (utils.py:112: silu / 0.596, of the
   * ROUNDED xo).  The decoder Block of an evaluation (networks_edm2.py:230 mp_cat, :73 mp_silu, :85 conv_skip): one launch
   * instead of the activation pass + the 1x1 conv, 13 of an evaluation's 141.  x2 == NULL: off.  A launch with x2 that the
   * LDS-DMA 1x1 kernel would take (>= 8192 positions) fails with ONIRIS_EUNSUPPORTED: the caller keeps the two launches.       */
  int32_t x_split;
  const void* x2;
  void* act_out;
  float cat_w1, cat_w2;
} OnirisConvArgs;

int oniris_conv_fwd(const OnirisConvArgs* args /* [host] */, oniris_stream_t stream);

/* Weight gradient of the same operator (replaces the autograd of F.conv2d / F.conv3d wrt the weight):
 *   slab[s][co][tap0+tap][ci] = sum_{(n,p) in split s} scale[n] * dy[n][p][co] * xframe(n)[p + tap][ci]
 * Split-K over position tiles: workgroup column s (gridDim.x = *nsplit_out <= nsplit_cap) owns slab s and writes
 * it with plain stores (no fp32 atomics: those run at ~1.3 TB/s chip-wide and dominated the first version).
 * path 0: xframe(n) = x[n] for all B*S*T frames.  path 1+j: frames n = (b,t), xframe = ctxframe(b, t+coff[j]),
 * dy = dy3 [B*T], tap0 = j*taps inside the 2*taps-deep slabs of the (2,3,3) weight.                              */
typedef struct OnirisWgradArgs {
  const void* x;          /* bf16 input frames   [B*xb_stride][H][W][Cin]                                         */
  const void* dy;         /* bf16 output grads   [B*T][H][W][Cout]                                                */
  void* dwp;              /* bf16 slabs [nsplit_cap][CoutP][taps_total][CinP]                                      */
  const float* scale;     /* [B*T] or NULL                                                                        */
  int32_t B, T, H, W, Cin, CinP, Cout, CoutP, taps;
  int32_t xb_stride, x_T, coff;   /* xframe(b,t) = x[b*xb_stride + t + coff] if 0 <= t+coff < x_T else fill       */
  float fill;
  int32_t nsplit_cap, taps_total, tap0;
  int32_t pad_;           /* variant: >= 0 LDS-DMA kernel where the shape allows it, < 0 register-staged kernel only        */
  int32_t* nsplit_out;    /* device int receiving the number of slabs written (= OnirisWeightDesc.nsplit)          */
} OnirisWgradArgs;

int oniris_conv_wgrad(const OnirisWgradArgs* args /* [host] */, oniris_stream_t stream);
/* 1..3 weight-gradient problems of identical geometry (H, W, channels, taps) in ONE launch -- the own-frame weight and
 * the two context taps of a gated conv.  The split-K workgroup columns are shared out in proportion to the groups'
 * position counts, so the launch writes (and oniris_weight_bwd later reads) a third of the slab bytes of three
 * separate launches.  Groups that address the same weight must use disjoint tap ranges (tap0).                    */
int oniris_conv_wgrad_group(const OnirisWgradArgs* args /* [host], ngroups entries */, int ngroups,
                            oniris_stream_t stream);

/* Backward pre-pass of the gated conv (autograd of edm2/conv.py:90-95): one pass over dout computing, per
 * frame-slot n, d_coef_own[n] = sum(dout*y2) (recovered as (sum(dout*out) - coef_ctx*sum(dout*y3)) / coef_own),
 * d_coef_ctx[n] = sum(dout*y3) (the gate gradient, chained to the 6 gating parameters on the host) and the
 * context-path gradient dy3[b,t] = sum_s coef_ctx[b,s,t] * dout[b,s,t].
 * dout/out bf16 [B][S][T][frame_elems], y3/dy3 [B][T][frame_elems].  d_coef_own / d_coef_ctx are ACCUMULATED (the launch is
 * split over pixel slices that meet through fp32 atomics): zero them before the call. */
int oniris_gconv_bwd_prep(const void* dout, const void* out, const void* y3, const float* coef_own,
                          const float* coef_ctx, float* d_coef_own, float* d_coef_ctx, void* dy3, int B, int S, int T,
                          int64_t frame_elems, oniris_stream_t stream);

/* The same pre-pass fused with the adjoint of the conv epilogue (the upstream gradient is read once):
 * mode 1 = ONIRIS_EPI_EMB_SILU (g = d u; raw = y; outputs dout = d y and d_cscale[n][co]),
 * mode 2 = ONIRIS_EPI_MPSUM    (g = d out; raw = v; xo = the clipped output, needed when clip > 0; outputs dres, dout = d v).
 * S = 2 (DART training layout), P = H*W pixels per frame, C channels (C % 8 == 0, C <= 512).
 * d_coef_own / d_coef_ctx / d_cscale are ACCUMULATED (pixel slices meet through atomics): zero them first.        */
int oniris_gconv_bwd_fused(int mode, const void* g, const void* raw, const void* y3, const float* coef_own,
                           const float* coef_ctx, const float* cscale, const void* xo, void* dout, void* dres, void* dy3,
                           float* d_coef_own, float* d_coef_ctx, float* d_cscale, int B, int T, int P, int C, float ta,
                           float tb, float clip,
                           int cscale_pitch /* floats between rows of cscale; 0 = C */,
                           const int32_t* clip_flag, float* coef_own_scaled, oniris_stream_t stream);
/* ABI 12: coef_own_scaled alone selects the protocol; clip_flag may be NULL when clip <= 0 (nothing to mask, ever), and dres may
 * be NULL (the residual's consumer reads g with the scale ta: oniris_act_bwd's dxo_scale).
 * clip_flag / coef_own_scaled (mode 2 only; ABI 10): the ALIASING protocol.  dout = tb * g * mask is a scaled
 * copy of the incoming gradient, so the launch does not write it: dgrad and weight gradient read g itself with the
 * coefficient vector coef_own_scaled[n] = tb * coef_own[n] (written here, [B*2*T]); `dout` is ignored (pass g).  When
 * *clip_flag != 0 (the forward clipped something: OnirisConvArgs.clip_flag) the mask is applied to g IN PLACE -- g must then
 * be a buffer this backward owns -- and xo is read; otherwise xo is not touched: 4 instead of 6 tensor passes.       */

/* ---------------------------------------------------------------------------------------------------------------
 * Fused magnitude-preserving glue (HBM-bound, one pass each) -- the elementwise chains of Block.forward
 * (edm2/networks_edm2.py:62-94) and their adjoints.  All tensors bf16 channels-last.
 * oniris_act_fwd: v = concat(w1*x[C1], w2*skip[C2]) (mp_cat, utils.py:128-134; C2 = 0: none); norm != 0: pixel norm
 *   v /= eps + |v|/sqrt(C) (utils.py:83-88), sden[pixel] receives the denominator; xo (optional) = v;
 *   a = silu(v)/0.596 (utils.py:112).   oniris_act_bwd: given da (and optionally dxo) -> dx [C1], dskip [C2]; dadd
 *   (optional, [C1]) is added to dx: a second, already complete gradient of x (an encoder output that is also a skip
 *   connection, networks_edm2.py:227-230) joins here instead of in a separate pass over three tensors.
 * oniris_emb_silu_bwd: backward of u = silu(y*c[n][co])/0.596: dy, and dc[n][co] = sum_pixels (fp32, overwritten).
 * oniris_mpsum_bwd: backward of out = clip(ta*res + tb*v): dres, dv (clip <= 0: no mask, `out` may be NULL).
 * oniris_resample: mode 0 = 2x2 mean (H,W = input size), mode 1 = nearest x2; result * scale (+ add, optional, shaped
 *   like out: see dadd above)  (utils.py:94-107 with f = [1,1]; adjoints: down^T = up * 0.25, up^T = down * 4).   */
int oniris_act_fwd(const void* x, const void* skip, void* xo, void* a, float* sden, int64_t npix, int C1, int C2,
                   float w1, float w2, int norm, int resample, int Ho, int Wo, oniris_stream_t stream);
/*   resample != 0: x is resampled on the way in (Block.forward's first line, networks_edm2.py:63): 1 = 2x2 mean, 2 = nearest
 *   x2, rounded to bf16 like oniris_resample stores it; npix and Ho x Wo describe the OUTPUT grid.                      */
int oniris_act_bwd(const void* da, const void* dxo, const void* xo, const float* sden, void* dx, void* dskip,
                   const void* dadd, int64_t npix, int C1, int C2, float w1, float w2, int norm,
                   float dxo_scale /* ABI 12: dxo enters as dxo_scale * dxo (1: as before) -- the residual gradient of an mp_sum
                                    * epilogue is ta times the gradient of its output: the caller hands that gradient itself */,
                   oniris_stream_t stream);
int oniris_emb_silu_bwd(const void* du, const void* y, const float* c, void* dy, float* dc, int N, int P, int C,
                        int c_pitch /* floats between rows of c; 0 = C */,
                        int dc_is_zero /* != 0: the caller hands in a zero-filled dc (no fill launch here) */,
                        oniris_stream_t stream);
int oniris_mpsum_bwd(const void* g, const void* out, void* dres, void* dv /* ABI 12: may be NULL (only dres is wanted) */,
                     int64_t numel, float ta, float tb, float clip, oniris_stream_t stream);
/* ABI 12, plain convs with the mp_sum + clip epilogue under the aliasing protocol (see oniris_gconv_bwd_fused): dv and dres are
 * scaled copies of g -- nothing is written, the consumers read g with tb / ta -- unless *clip_flag != 0 (OnirisConvArgs.clip_flag of
 * the forward launch): then g is masked IN PLACE where |out| reached the clip.                                          */
int oniris_mpsum_mask(void* g, const void* out, int64_t numel, float clip, const int32_t* clip_flag, oniris_stream_t stream);
int oniris_resample(const void* in, void* out, const void* add, int64_t N, int H, int W, int C, int mode, float scale,
                    oniris_stream_t stream);
/* ABI 12: the same with a general separable filter (utils.py:94-107: `taps` [host] = the 1-D filter f NORMALISED to sum 1, an even
 * number of taps 2 .. 8, padding (ntaps - 1) / 2): mode 0 = depthwise conv2d with outer(f, f), stride 2; mode 1 = depthwise
 * conv_transpose2d with 4 * outer(f, f), stride 2.  taps = {0.5, 0.5} is oniris_resample.  Block(resample_filter=...),
 * networks_edm2.py:26,66.                                                                                               */
int oniris_resample_filter(const void* in, void* out, const void* add, int64_t N, int H, int W, int C, int mode,
                           const float* taps /* [host] */, int ntaps, float scale, oniris_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * VideoAttention / FrameAttention (edm2/attention/attention_modules.py:30-82, 105-119; RoPe.py:43-68).
 *
 * oniris_qkv_norm: splits the 1x1-conv output qkv [N][P][3C] (channel = s*C + head*64 + c, see perm3) into
 * q,k,v [N][P][C], each normalised per token and head over its 64 channels (normalize(dim=-1),
 * attention_modules.py:38,49); q additionally carries the softmax scale, q' = log2(e)/sqrt(64) * q: the attention
 * entry points below expect that (their scores are log2-domain straight out of the MFMA) and return dq with respect to
 * the UNSCALED normalised q, which is what oniris_qkv_norm_bwd (the adjoint: dq,dk,dv -> dqkv) takes.
 * oniris_rope: rotates q (mode 1: * scale) or k (mode 2: / scale) over the FRAME index with host-built fp32
 * tables cos/sin/scale [n_pos][64] (built from fp16-rounded angles exactly like RoPe.py:21-32), position of
 * frame f = pos_offset + (f % pos_mod); writes the rotated tensor [B][L][C] and/or the transposed copy
 * [B][heads][64][L] (the k-contiguous operand layout of the PV / dK / dQ MFMA products).  mode 0: copy only.
 * mode 3/4: adjoint of mode 1/2 (backward).
 */
/* kv_tokens_per_batch > 0: k and v are written into a KV ring (edm2/sampler.py rollout): token r of batch b goes to
 * element b * kv_batch_stride + (kv_token_offset + r) * C of k / v; 0: dense [n_tokens][C] like q.                 */
int oniris_qkv_norm(const void* qkv, void* q, void* k, void* v, int64_t n_tokens, int C, int64_t kv_tokens_per_batch,
                    int64_t kv_batch_stride, int64_t kv_token_offset, oniris_stream_t stream);
int oniris_qkv_norm_bwd(const void* qkv, const void* dq, const void* dk, const void* dv, void* dqkv,
                        int64_t n_tokens, int C, oniris_stream_t stream);
/* The same with the rotary embedding of q and k fused in (VideoAttention in training, RoPe.py:43-68: position of a token
 * = (token / P) mod pos_mod; cos / sin / scale tables [pos][64] fp32): q, k leave rotated (q also carries the softmax
 * scale), one bf16 rounding; the _bwd takes the attention backward's dq, dk, dv and returns dqkv.                       */
int oniris_qkv_norm_rope(const void* qkv, void* q, void* k, void* v, const float* cos_t, const float* sin_t,
                         const float* scale_t, int64_t n_tokens, int C, int P, int pos_mod, oniris_stream_t stream);
int oniris_qkv_norm_rope_bwd(const void* qkv, const void* dq, const void* dk, const void* dv, void* dqkv,
                             const float* cos_t, const float* sin_t, const float* scale_t, int64_t n_tokens, int C, int P,
                             int pos_mod, oniris_stream_t stream);
int oniris_rope(const void* x, void* xr, void* xt, const float* cos_t, const float* sin_t, const float* scale_t,
                int mode, int B, int frames, int P, int C, int pos_offset, int pos_mod,
                int64_t x_batch_stride /* elements between the sequences of x (KV ring); 0 = frames*P*C */,
                int64_t xr_batch_stride /* the same for xr */, oniris_stream_t stream);
/* Heads of 8, 16 or 32 channels (Block(channels_per_head=...), networks_edm2.py:28,39; the reference's own tests use 16,
 * consistency_test.py:39,61).  The attention entry points below are written for 64-channel heads; other sizes run through
 * them PADDED: q, k, v [tokens][heads*64] with channels head_dim..63 of every head zero (q.k and P.V are unchanged by
 * zeros), the softmax scale log2(e)/sqrt(head_dim) on q.  These three do what depends on head_dim: the per-head pixel
 * norm (attention_modules.py:48-49), the rotary embedding with its partner head_dim/2 channels away (RoPe.py:34-57; tables
 * [pos][head_dim] fp32, position of a token = ((token / P) % seq_frames + pos_off) % pos_mod with seq_frames = frames per
 * sequence of the tensor; rope bit 0: rotate q, bit 1: rotate k) and the
 * adjoint (dq as the attention backward returns it for the 64-channel model, dk, dv padded -> dqkv).
 * qkv / dqkv [tokens][3*heads*head_dim], channel = (s*heads + head)*head_dim + c.  oniris_rope_hd: mode 1 = q (* scale),
 * 2 = k (/ scale) on a padded tensor (eval: all cached keys are re-rotated per call, RoPe.py:55-57).               */
int oniris_qkv_norm_hd(const void* qkv, void* q, void* k, void* v, const float* cos_t, const float* sin_t,
                       const float* scale_t, int64_t n_tokens, int heads, int head_dim, int P, int pos_mod, int pos_off,
                       int rope, int seq_frames, oniris_stream_t stream);
int oniris_qkv_norm_hd_bwd(const void* qkv, const void* dq, const void* dk, const void* dv, void* dqkv, const float* cos_t,
                           const float* sin_t, const float* scale_t, int64_t n_tokens, int heads, int head_dim, int P,
                           int pos_mod, int pos_off, int rope, int seq_frames, oniris_stream_t stream);
int oniris_rope_hd(const void* x, void* out, const float* cos_t, const float* sin_t, const float* scale_t, int64_t n_tokens,
                   int heads, int head_dim, int P, int pos_mod, int pos_off, int mode, int seq_frames, oniris_stream_t stream);
/* One new frame per sequence in the KV-cached sampler (edm2/sampler.py:12-85; attention_modules.py:51-70): oniris_qkv_norm
 * + the rotation of the frame's q and k at table row `pos` (= number of keys - 1) in one pass.  k (un-rotated) and v go
 * into the KV ring as in oniris_qkv_norm; kr receives the ROTATED k at the same ring position: the ring's rotated image,
 * whose committed frames the host rotates once per frame count (oniris_rope mode 2 with xr_batch_stride) instead of once
 * per UNet evaluation (31 per generated frame).                                                                     */
int oniris_qkv_norm_rope_eval(const void* qkv, void* q, void* k, void* v, void* kr, const float* cos_t, const float* sin_t,
                              const float* scale_t, int64_t n_tokens, int C, int64_t kv_tokens_per_batch,
                              int64_t kv_batch_stride, int64_t kv_token_offset, int pos, oniris_stream_t stream);

/* oniris_qkv_eval: the attn_qkv 1x1 convolution + oniris_qkv_norm (tables NULL; kr NULL; kv_tokens_per_batch may be 0 =
 * dense k, v) or + oniris_qkv_norm_rope_eval (tables given) in ONE launch, for the sampler's evaluations: x [n_tokens][C]
 * bf16 channels-last, w = the packed forward weight of attn_qkv [3C][CinP] bf16 (rows (s m c), see perm3); outputs as those
 * entry points write them (attention_modules.py:47-57).                                                               */
int oniris_qkv_eval(const void* x, const void* w, void* q, void* k, void* v, void* kr, const float* cos_t, const float* sin_t,
                    const float* scale_t, int64_t n_tokens, int C, int CinP, int64_t kv_tokens_per_batch,
                    int64_t kv_batch_stride, int64_t kv_token_offset, int pos, oniris_stream_t stream);

/* Block-sparse flash attention forward (replaces compiled_flex_attention / F.scaled_dot_product_attention,
 * attention_modules.py:41,66,70,75,115).  q [B][Lq][C], k,v [B][Lk][C] bf16 (head h = channels 64h..64h+63),
 * (the transposed operands V^T, K^T, Q^T, dO^T are produced inside the kernels by transposing LDS reads; the
 * qt/kt/vt/doutt fields are reserved and may be NULL).  softmax scale 1/sqrt(64).
 *   mask_mode 0: dense;  1: frame-causal (key frame <= query frame, frames of P tokens; query frames are the
 *   LAST Lq/P frames of the key sequence);  2: DART training mask (mask_mod of TrainingMask, T frames per half).
 *   kv_num/kv_idx: device int32 table of 128-token blocks for one (b,h) ([nrows], [nrows][ncols]); NULL = every
 *   block (then mask_mode alone decides).  The kernel visits exactly the listed blocks and applies mask_mod per
 *   element -- the semantics of the compiled FlexAttention kernel.
 * out [B][Lq][C] bf16, lse [B][heads][Lq] fp32 (log2-domain, for the backward).                                 */
typedef struct OnirisAttnArgs {
  const void *q, *k, *v, *qt, *kt, *vt;   /* qt/kt only needed by the backward                                    */
  void* out;
  float* lse;
  const int32_t *kv_num, *kv_idx;         /* forward / dQ table (rows = q blocks)                                 */
  const int32_t *q_num, *q_idx;           /* transposed table (rows = kv blocks), backward only                   */
  int32_t tab_cols, qtab_cols;            /* row pitch of kv_idx / q_idx                                          */
  int32_t B, heads, Lq, Lk, C;
  int32_t mask_mode, P, T;
  int32_t tab_block;                      /* tokens per table block (BlockMask BLOCK_SIZE: 128, or P if P >= 128)   */
  /* backward */
  const void *dout, *doutt;               /* bf16 [B][Lq][C], [B][heads][64][Lq]                                  */
  const float* delta;                     /* [B][heads][Lq]                                                       */
  void *dq, *dk, *dv;                     /* bf16 [B][L][C]                                                       */
  /* dK/dV load balancing (oniris_attn_bwd_dkv): with a causal table the first key blocks are attended by every later
   * query, the last by almost none, and a key block is one workgroup -- the launch takes as long as its longest
   * query list.  dkv_chunks > 1 splits every list into that many contiguous chunks, one workgroup each, which write
   * fp32 partial sums to dkv_part [2 (dk|dv)][chunks][B][Lk][C]; a second kernel adds them in chunk order
   * (deterministic, no atomics) and writes the bf16 dk, dv.  0 / 1: one workgroup per key block, no scratch.       */
  float* dkv_part;
  /* dkv_item_keys (scheduled dK/dV launch only; ABI 12, was padding): keys per work item of `sched` -- 0 / 64: half a 128-token
   * table block per item (two query halves per key half: the two halves of a workgroup meet through LDS at the end), 128: a
   * whole table block per item, every compute wave owns 32 keys against all 128 rows of a query block -- half the LDS-DMA
   * bytes and instructions per MFMA, no merge; needs enough items per workgroup for the heaviest one not to dominate
   * (the caller decides from the schedule's weights).                                                                  */
  int32_t dkv_chunks, dkv_item_keys;
  /* static balanced schedule for the persistent kernels (oniris_attn_schedule): device int32 [sched_wgs][sched_slots],
   * entry = (pair << 16) | block with pair = b * heads + head, or -1.  NULL: one workgroup per block (grid kernels).   */
  const int32_t* sched;
  int32_t sched_wgs, sched_slots;
  int64_t v_bstride;                      /* elements between the sequences of v (a KV ring); 0 = Lk*C (forward only) */
  int64_t k_bstride;                      /* the same for k (the ring's rotated image)                                */
  /* split-KV decode (oniris_attn_fwd, mask_mode 0, no schedule): kv_splits > 1 deals the key tiles of every query block to
   * that many workgroups, which leave un-normalised partials (O, l) in split_ws [kv_splits][B][heads][Lq][65] fp32; a
   * second kernel adds them and normalises (one new frame against a long KV ring would otherwise run on heads * B CUs)  */
  float* split_ws;
  /* frame_kernel (was padding): dense attention inside frames of 64 / 128 / 256 tokens (mask_mode 0, Lq == Lk, no table, no ring
   * strides, no split) runs on kernels of its own (csrc/attention_frame.h: one workgroup per 256 consecutive tokens of a head,
   * K | V staged once); bit 0 set = keep such launches on the generic grid kernels (A/B, tests).  Bit 1 (round 6): a mask_mode 0
   * launch of at most 512 32-row query blocks against Lk >= 256 keys (one new frame against a KV ring below the split-KV
   * threshold) runs four key streams per workgroup; Lk > Lq; bit 1 set = keep it on the one-stream kernel (A/B, tests).  Bit 2 (round 6): a
   * forward launch over at most 64 (frame, head) pairs of 256 tokens runs two query halves per frame; bit 2 set = one workgroup per pair.   */
  int32_t kv_splits, frame_kernel;
} OnirisAttnArgs;

/* Dense attention inside frames of 64 / 128 / 256 tokens (FrameAttention.forward, attention_modules.py:105-119; VideoAttention's
 * just_2d branch, :36-45): the WHOLE backward in one launch -- delta = dout . out, dq, dk, dv -- reading q, k, v, out, dout once
 * (the op is HBM-bound: 128 FLOP per byte at 256 tokens per frame).  Takes the forward's arguments (mask_mode 0, Lq == Lk == P,
 * lse from oniris_attn_fwd, plain -- not negated) plus dout, dq, dk, dv; no delta, no scratch.  csrc/attention_frame.h.          */
int oniris_frame_attn_bwd(const OnirisAttnArgs* args, oniris_stream_t stream);
/* The same layer core straight from the attn_qkv output (FrameAttention.forward, attention_modules.py:108-115: rearrange, normalize(dim=-1),
 * scaled_dot_product_attention): qkv [n_frames * P][3 C] bf16 with channel = s * C + head * 64 + c (the packed attn_qkv order), C =
 * heads * 64, P in {64, 128, 256}.  The per-head normalisation of q, k, v happens inside (registers / LDS), the backward applies its
 * adjoint to the fp32 dq, dk, dv and writes dqkv [n_frames * P][3 C]: no q / k / v / dq / dk / dv tensors exist (the normalisation
 * passes were half of the layer's HBM traffic).  out [n_frames * P][C] bf16, lse [n_frames][heads][P] fp32 (log2 domain).          */
int oniris_frame_attn_qkv_fwd(const void* qkv, void* out, float* lse, int64_t n_frames, int P, int heads, oniris_stream_t stream);
int oniris_frame_attn_qkv_bwd(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv, int64_t n_frames,
                              int P, int heads, oniris_stream_t stream);

/* Static load balancing of block-sparse attention [host]: n_pairs (batch, head) pairs x n_blocks work items per pair
 * (query blocks for the forward / dQ, key blocks for dK/dV), weight[blk] = cost of block blk (its list length in the
 * mask table + a fixed per-item cost; the same for every pair).  The n_wg persistent workgroups form n_groups =
 * 8 / 4 / 2 / 1 groups (workgroup w -> group w % n_groups: the XCD it is dispatched to under round-robin placement --
 * speed only, never correctness); pair p belongs to group p % n_groups, so that its K / V stay in one L2.  Inside a
 * group the items are dealt longest-processing-time first to the least loaded workgroup.  Writes
 * sched[n_wg][n_slots] (see OnirisAttnArgs.sched) and returns the number of slots used (<= n_slots); with sched ==
 * NULL only returns the number of slots needed.  <0 on error.                                                       */
int oniris_attn_schedule(int n_pairs, int n_blocks, const int32_t* weight, int n_wg, int32_t* sched, int n_slots);

int oniris_attn_fwd(const OnirisAttnArgs* args /* [host] */, oniris_stream_t stream);
/* delta[b][h][q] = sum_c dout*out ; doutt (optional) = transposed dout; neg (optional, needs lse) [2][B][heads][L] =
 * -lse | -delta: the row constants the scheduled dQ and dK/dV kernels start their S / dP chains from (with
 * OnirisAttnArgs.sched set, .lse / .delta of oniris_attn_bwd_dq and oniris_attn_bwd_dkv point at these two planes; the dQ
 * kernel takes the FORWARD's work list -- 128-row query blocks --, the dK/dV kernel the one over 64-key items)       */
int oniris_attn_bwd_prep(const void* dout, const void* out, float* delta, void* doutt, const float* lse, float* neg, int B,
                         int heads, int L, int C, oniris_stream_t stream);
int oniris_attn_bwd_dq(const OnirisAttnArgs* args /* [host] */, oniris_stream_t stream);
int oniris_attn_bwd_dkv(const OnirisAttnArgs* args /* [host] */, oniris_stream_t stream);

/* The gradient exchange of the data-parallel loop (cs_train.py:53-54,108-114,168) is issued by the host through
 * torch.distributed ("nccl" == RCCL on ROCm) on the flat gradient buffer: autoregressive_diffusion_amd/parallel.py.
 * This library exports no collective wrappers (the four pass-through oniris_comm_* entry points of ABI <= 7 carried no
 * logic and are gone since ABI 8).                                                                                    */

/* ---------------------------------------------------------------------------------------------------------------
 * fp32 verification path -- Precond(use_fp16=False) / Precond.forward(force_fp32=True) (networks_edm2.py:285,294: the switch
 * that picks the arithmetic type of the whole net; the reference's modules compute in the dtype of their input, conv.py:37-46).
 * Activations, weights, products and sums in fp32 (contractions on v_mfma_f32_32x32x2_f32); channels-last fp32 tensors
 * [N][H][W][C] with ANY channel counts.  Not the timed path: it exists so that the reference's criterion std(diff) <= 3e-4
 * (edm2/consistency_test.py:23-32) can be held against the fp32 fixtures.  csrc/fp32.hip.
 * oniris_conv_f32: out[n][y][x][co] = sum_{tap, ci} w[tap][co][ci] * x[n][y + dy][x + dx][ci], taps = 1 (1x1 / linear) or 9 (3x3,
 * tap = 3 (dy + 1) + (dx + 1), zero padding) -- F.conv2d of MPConv.forward (conv.py:41-46); the data gradient is the same call
 * on w'[8 - tap][ci][co].  oniris_wgrad_f32: dw[tap][co][ci] += sum_pos dy[pos][co] * x[shift(pos, tap)][ci] (dw zeroed by the
 * caller; fp32 atomics over position chunks).                                                                              */
int oniris_conv_f32(const float* x, const float* w, float* out, int64_t N, int H, int W, int Cin, int Cout, int taps,
                    oniris_stream_t stream);
int oniris_wgrad_f32(const float* x, const float* dy, float* dw, int64_t N, int H, int W, int Cin, int Cout, int taps,
                     oniris_stream_t stream);
/* Softmax attention in fp32 for any head width D <= 256 (attention_modules.py:59-77,105-119; also what serves heads wider than
 * the 64 channels of the product kernels, networks_edm2.py:28,39).  q [BH][Lq][D], k / v [BH][Lk][D], out [BH][Lq][D],
 * lse [BH][Lq] (natural log), all fp32 contiguous; logits = scale * q.k.
 * mask_mode 0: dense.  1: frame-causal, key frame <= query frame + q_frame_off (frames of P tokens; q_frame_off = frames
 * already cached: causal prefill and cached steps, attention_modules.py:69-77).  2: the DART training mask over 2T frames
 * (clean | noised) = BlockMask table AND mask_mod as the compiled FlexAttention evaluates it (attention_masking.py:27-53).
 * _bwd: dq, dk, dv from dout, lse and delta[i] = sum_c dout[i][c] * out[i][c].                                              */
typedef struct OnirisAttnF32Args {
  const float* q; const float* k; const float* v; float* out; float* lse;
  const float* dout; const float* delta; float* dq; float* dk; float* dv;
  int32_t BH, Lq, Lk, D, mask_mode, P, T, q_frame_off;
  float scale; int32_t pad_;
} OnirisAttnF32Args;
int oniris_attn_f32_fwd(const OnirisAttnF32Args* args, oniris_stream_t stream);
int oniris_attn_f32_bwd(const OnirisAttnF32Args* args, oniris_stream_t stream);

#ifdef __cplusplus
}
#endif

#endif
