/*
 * rip_hip.h — C ABI of librip_hip.so: the MI355X (gfx950) implementation of
 * oatomobile's deep-imitative-model inference path.
 *
 * The reference has no FFI for this path: it is a Python class API
 *   ImitativeModel  oatomobile/baselines/torch/dim/model.py:36-253
 *   RIPAgent        oatomobile/baselines/torch/rip/agent.py:30-151
 *   AutoregressiveFlow oatomobile/torch/networks/sequence.py:28-216
 * whose tensor math dispatches to ATen.  Each entry point below replaces the
 * ATen work of one reference method (cited per function) and is what a
 * maintainer binds with ctypes.  oatomobile_amd/_lib.py is
 * that binding.
 *
 * Conventions
 *   - return 0 on success, a negative RIP_E* code otherwise; never throws.
 *     rip_last_error() returns a thread-local message for the last failure.
 *   - every `*_dev` pointer is caller-owned device memory (e.g. a PyTorch-ROCm
 *     tensor's data_ptr()), contiguous fp32 unless stated otherwise.
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and
 *     the call returns without synchronising: every entry point that takes a
 *     stream only validates arguments and launches kernels — no hipMalloc /
 *     hipFree / hipMemcpy / device synchronisation (safe under stream capture).
 *     All device scratch is allocated by rip_create from (max_batch,
 *     max_candidates); a call that exceeds it returns RIP_ESTATE.
 *     rip_create / rip_destroy / rip_load_model / rip_train_create / _destroy
 *     are the only ones that allocate or copy synchronously.  NULL = the null stream.
 *   - a handle is bound to one device: every entry point that takes a handle
 *     makes that device current for the duration of the call and restores the
 *     caller's current device before returning.  The stateless entry points
 *     (rip_transform, rip_goal_likelihood, rip_lidar_bev, rip_cil_decode) launch
 *     on the caller's current device, which must own the pointers.
 *   - a handle's scratch is shared by its calls, so a handle is single-stream
 *     and not thread-safe: when consecutive calls on one handle name different
 *     streams, the later stream is made to wait (event) for the earlier one.
 *   - trajectory shape is fixed at T=4 steps x D=2, hidden size 64, like the
 *     reference's ImitativeModel(output_shape=(4, 2)) (dim/model.py:41-68).
 */
#ifndef RIP_HIP_H_
#define RIP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rip_handle rip_handle;
typedef void* rip_stream_t;

enum {
  RIP_OK = 0,
  RIP_EINVAL = -1,  /* bad argument (shape, NULL, range) */
  RIP_EHIP = -2,    /* a HIP runtime call failed */
  RIP_ESTATE = -3   /* model k not loaded yet, scratch sized by rip_create too small, ... */
};

/* rip/agent.py:121-127 as coded: "WCM" = min_k(-posterior), "BCM" = max_k, "MA" = mean_k. */
enum { RIP_ALGO_WCM = 0, RIP_ALGO_MA = 1, RIP_ALGO_BCM = 2 };
/* encoder arithmetic: fp32 everywhere, or bf16 weights+activations with fp32 accumulate. */
enum { RIP_ENC_FP32 = 0, RIP_ENC_BF16 = 1 };

#define RIP_T 4
#define RIP_D 2
#define RIP_HIDDEN 64
#define RIP_MAX_MODELS 8
#define RIP_MAX_STEPS 64

/* ABI version of this header (bumped on any signature change). */
int rip_abi_version(void);
const char* rip_last_error(void);

/* Replaces ImitativeModel.__init__/.to(device) for K ensemble members
 * (dim/model.py:36-74; rip/agent.py:49-50).  Allocates, on HIP device `device`,
 * the weights of K models with `in_channels` BEV channels and ALL scratch any
 * later call may use: the encoder workspace for up to `max_batch` observations
 * per call and the plan-search scratch (candidate plans, best losses, adjoint
 * tape, ImitativeModel.forward traces) for up to `max_batch` x `max_candidates`
 * latent rows.  Later calls never allocate. */
int rip_create(rip_handle** out, int K, int in_channels, int max_batch, int max_candidates, int device);
int rip_destroy(rip_handle* h);

/* Replaces model.load_state_dict(torch.load(ckpt)) (README.md:57-58,
 * torch/savers.py:45-55).  `packed_host` = the fp32 tensors of the reference
 * state_dict in its own key order minus the int64 num_batches_tracked counters
 * (oatomobile_amd/arch.py:packed_spec), `numel` floats.  BatchNorm (eps 1e-5,
 * running statistics = eval mode) is folded into the preceding conv here. */
int rip_load_model(rip_handle* h, int k, const float* packed_host, size_t numel);

/* R2 — ImitativeModel.transform on `lidar` (dim/model.py:245-251 ->
 * torch/transforms.py:34-49): bilinear (H,W)->(out_hw,out_hw) with
 * align_corners=True, then swap H and W.  in: [B,C,H,W] (channels_last=0) or
 * [B,H,W,C] (channels_last=1, the sensor layout rip/agent.py:69 transposes on
 * the host).  out: [B,C,out_hw,out_hw] NCHW fp32. */
int rip_transform(const float* lidar_dev, int B, int C, int H, int W, int channels_last, int out_hw,
                  float* out_dev, rip_stream_t stream);

/* R3+R4 — ImitativeModel._params for models [k_begin, k_begin+k_count)
 * (dim/model.py:173-219): MobileNetV2 encoder -> cat(feat128, vec5) -> merger.
 * visual_dev [B,C,100,100] (output of rip_transform), vec_dev [B,5] =
 * (velocity[3], is_at_traffic_light, traffic_light_state).
 * z_dev [k_count,B,64]; feat_dev optional [k_count,B,128] encoder logits (NULL to skip).
 * enc_dtype: RIP_ENC_FP32 | RIP_ENC_BF16. */
int rip_encode(rip_handle* h, const float* visual_dev, const float* vec_dev, int B, int k_begin, int k_count,
               int enc_dtype, float* z_dev, float* feat_dev, rip_stream_t stream);

/* Diagnostics tap (tests: every conv layer of the encoder against the oracle, teacher-forced).  Runs the encoders of
 * models [k_begin, k_begin + k_count)
 * on B observations with the handle's CURRENT kernel selection — the one rip_encode makes for the same (B, k_count):
 * the bf16 selection keys on B * k_count (launch shapes, the tile blocks from 64 pairs, the row-streaming depthwise and
 * persistent GEMM kernels of large launches), so a test of the kernels a K-model call ships must tap with that k_count —
 * up to and including conv layer `layer` (network order of
 * torchvision's `features`: 0 = features.0, then [expand,] depthwise, project of features.1..17, 51 = features.18 — the
 * reference builds it at torch/networks/perception.py:36-51) and writes that layer's output as fp32:
 * `dst_dev` [k_count][B][H][W][C] (NHWC; bf16 activations are widened exactly), or [k_count][B][1280] for layer 51, whose 4x4 average
 * pool is part of its epilogue.  `dst_numel` must be exactly that size — or `dst_dev` = NULL with `dst_numel` = 0: the launch
 * sequence runs up to the layer and nothing is copied out (bench.py times the encoder block by block that way).  Overwrites the handle's activation workspace;
 * z is not produced.  RIP_EINVAL when the layer's output never reaches memory under the current
 * RIP_OPT_ENCODER_FUSED setting (an interior layer of a fused block). */
int rip_encode_tap_k(rip_handle* h, const float* visual_dev, int B, int k_begin, int k_count, int enc_dtype, int layer,
                     float* dst_dev, size_t dst_numel, rip_stream_t stream);
/* The same for ONE model: rip_encode_tap_k(h, visual, B, k, 1, ...). */
int rip_encode_tap(rip_handle* h, const float* visual_dev, int B, int k, int enc_dtype, int layer, float* dst_dev,
                   size_t dst_numel, rip_stream_t stream);

/* Fused R2+R3+R4 for the agent's hot loop: raw sensor BEV [B,H,W,C]
 * (channels_last=1) or [B,C,H,W] -> z (any H, W >= 1 like F.interpolate; the
 * CARLA sensor gives 200 x 200).  Same results as rip_transform + rip_encode. */
int rip_encode_raw(rip_handle* h, const float* lidar_dev, int channels_last, int H, int W, const float* vec_dev,
                   int B, int k_begin, int k_count, int enc_dtype, float* z_dev, rip_stream_t stream);

/* rip_encode_raw on a CODED BEV (the replay cache of oatomobile_amd/replay.py): codes_dev
 * [B,H,W,C] uint8 (sensor layout), every cell an index into lut_dev [256] float32 = the distinct values of the float32
 * BEV it was packed from (the CARLA LIDAR histogram has six: k/5, utils/carla.py:225-233).  The table is applied while
 * the transform stages its input, so z is bit-identical to rip_encode_raw on the float32 BEV at a quarter of the
 * H2D / HBM bytes.  C <= 3 and a down-sampling factor <= 2 (200 x 200 -> 100 x 100). */
int rip_encode_raw_u8(rip_handle* h, const uint8_t* codes_dev, const float* lut_dev, int H, int W, const float* vec_dev,
                      int B, int k_begin, int k_count, int enc_dtype, float* z_dev, rip_stream_t stream);

/* R6 — AutoregressiveFlow._forward of model k (sequence.py:95-151).
 * x_dev [N,4,2]; z_dev [z_rows,64] with z_rows == N or 1 (broadcast);
 * y_dev [N,4,2]; logabsdet_dev [N] (NULL to skip). */
int rip_flow_forward(rip_handle* h, int k, const float* x_dev, const float* z_dev, int N, int z_rows,
                     float* y_dev, float* logabsdet_dev, rip_stream_t stream);

/* R7 — AutoregressiveFlow._inverse of model k (sequence.py:153-216).
 * y_dev [N,4,2] -> x_dev [N,4,2] (NULL to skip), log_prob_dev [N], logabsdet_dev [N]. */
int rip_flow_inverse(rip_handle* h, int k, const float* y_dev, const float* z_dev, int N, int z_rows,
                     float* x_dev, float* log_prob_dev, float* logabsdet_dev, rip_stream_t stream);

/* R9 — ImitativeModel._goal_likelihood per plan row, before its mean(dim=0)
 * (dim/model.py:143-171).  y_dev [N,4,2]; goal_dev [goal_rows,G,2] with
 * goal_rows == N or 1; rows_dev [N]. */
int rip_goal_likelihood(const float* y_dev, const float* goal_dev, int N, int goal_rows, int G, float epsilon,
                        float* rows_dev, rip_stream_t stream);

/* Scoring mode of R5 (rip/agent.py:109-119, per plan instead of batch mean):
 * S[k,b,n] = log_prob_k - logabsdet_k (+ goal log-likelihood if goal_dev != NULL)
 * for models [k_begin, k_begin+k_count).  z_dev [k_count,B,64]; y_dev [B,N,4,2];
 * goal_dev [B,G,2] or NULL; S_dev [k_count,B,N].  This [K,N] matrix is what the
 * multi-GPU all-gather carries. */
int rip_score(rip_handle* h, int k_begin, int k_count, const float* z_dev, const float* y_dev,
              const float* goal_dev, int B, int N, int G, float epsilon, float* S_dev, rip_stream_t stream);

/* Ensemble aggregation of a (gathered) score matrix, rip/agent.py:121-127 as coded but per plan:
 * S_dev [K,B,N] -> loss_dev [B,N] = WCM: min_k(-S), BCM: max_k(-S), MA: mean_k(-S) (NULL to skip) and
 * best_index_dev [B] int32 = argmin_n loss (NULL to skip).  In the model-parallel layout every rank calls this
 * on the all-gathered matrix, so no second collective is needed. */
int rip_aggregate_scores(const float* S_dev, int K, int B, int N, int algorithm, float* loss_dev,
                         int32_t* best_index_dev, rip_stream_t stream);

/* R5 — the RIPAgent.__call__ plan search (rip/agent.py:78-137) for B
 * observations x N candidate latents, all K loaded models.
 *   z_dev [K,B,64]; goal_dev [B,G,2]; x0_dev [B,N,4,2] (row n=0 zeros = the
 *   reference start, rip/agent.py:85-90).
 * Every candidate runs `num_steps` Adam(lr) steps on its own latent with its own
 * loss/x_best bookkeeping (post-step x against pre-step loss, :131-135); the
 * plan of the candidate with the lowest best-loss wins.  N=1 is the reference.
 * Outputs (any may be NULL): plan_dev [B,4,2]; plans_dev [B,N,4,2];
 * loss_best_dev [B,N]; best_index_dev [B] int32;
 * trace_post_dev [num_steps,K,B,N] per-step posteriors (incl. the goal term);
 * trace_x_dev [num_steps,B,N,4,2] post-step latents;
 * trace_grad_dev [num_steps,B,N,4,2] dLoss/dx of every step (what Adam consumes).
 * Both search kernels implement the traces (the MFMA-batched one for N % 32 == 0). */
int rip_search(rip_handle* h, const float* z_dev, const float* goal_dev, const float* x0_dev, int B, int N, int G,
               int algorithm, int num_steps, float lr, float epsilon, float* plan_dev, float* plans_dev,
               float* loss_best_dev, int32_t* best_index_dev, float* trace_post_dev, float* trace_x_dev,
               float* trace_grad_dev, rip_stream_t stream);

/* Gradient-mode model-parallel search (BASELINE configs[3]: the K models of the ensemble live on
 * different GPUs).  One Adam step of rip/agent.py:102-135 is cut where the ensemble is reduced; the caller moves
 * ONE [K_local,B,N,9] block per step between ranks (all-gather, e.g. torch.distributed over RCCL):
 *
 *   rip_mp_local: y = F(x; z_fwd) with the flow of handle model `k_fwd` (the ensemble's model 0, whose flow
 *     weights every rank holds), then for this rank's models [k_begin, k_begin+k_count): inverse(y; z_k) and its
 *     adjoint -> out_dev[k,b,n,:] = (q_k = log_prob - logabsdet, dq_k/dy[8]).  first_is_fwd != 0 (the rank that
 *     owns model 0): row 0 reports q_0 through the self-inverse shortcut with a zero gradient (it is folded into
 *     the forward adjoint of rip_mp_update).
 *   rip_mp_update: on the gathered [K,B,N,9] matrix (model 0 first), redundantly on every rank: ensemble
 *     aggregation (rip/agent.py:121-127), goal term, adjoint of F, Adam step `step` (0-based) and the
 *     loss_best / x_best bookkeeping (:131-135) on the rank-replicated state x, m, v, x_best [B,N,4,2] and
 *     loss_best [B,N] (initialise m = v = 0, x_best = x, loss_best = 1000).  grad_dev [B,N,4,2] optional.
 * One rank holding all K models reproduces rip_search's wave-per-chain kernel. */
int rip_mp_local(rip_handle* h, int k_fwd, int k_begin, int k_count, int first_is_fwd, const float* z_fwd_dev,
                 const float* z_dev, const float* x_dev, int B, int N, float* out_dev, rip_stream_t stream);
int rip_mp_update(rip_handle* h, int k_fwd, const float* z_fwd_dev, const float* gathered_dev, int K,
                  const float* goal_dev, int B, int N, int G, int algorithm, int step, float lr, float epsilon,
                  float* x_dev, float* m_dev, float* v_dev, float* x_best_dev, float* loss_best_dev, float* grad_dev,
                  rip_stream_t stream);

/* N5 — the sensor step in front of R2: carla_lidar_measurement_to_ndarray
 * (oatomobile/utils/carla.py:165-233) on already parsed point clouds.  points_dev [P_total,3] fp32 (x, y, z as in
 * LidarMeasurement.raw_data, :212-213); offsets_dev [B+1] int32, observation b owns points [offsets[b], offsets[b+1]);
 * bev_dev [B,200,200,2] fp32: per height channel (z <= -2.5 / z >= -2.5) the np.histogramdd counts over
 * np.linspace(-50, 51, 201)^2, clipped at 5 and divided by 5.  Bit-exact with the reference.  Stateless. */
int rip_lidar_bev(const float* points_dev, const int32_t* offsets_dev, int B, float* bev_dev, rip_stream_t stream);

/* N4 — BehaviouralModel.forward after the encoder (oatomobile/baselines/torch/cil/model.py:88-127):
 * merger MLP over cat(features, velocity, is_at_traffic_light, traffic_light_state, mode), then the GRUCell + Linear
 * residual rollout.  feat_dev [B,128] = MobileNetV2 logits (rip_encode's feat_dev of a handle loaded with the model's
 * encoder); vec_dev [B,6]; weights_dev = rip_cil_blob_floats() fp32 values in BehaviouralModel.state_dict() order
 * (_merger._model.{0,2,4}.{weight,bias}, _decoder.{weight_ih,weight_hh,bias_ih,bias_hh}, _output.{weight,bias});
 * y_dev [B,T,2] (T = 40 in the reference).  Stateless. */
int rip_cil_decode(const float* feat_dev, const float* vec_dev, const float* weights_dev, int B, int T, float* y_dev,
                   rip_stream_t stream);
int rip_cil_blob_floats(void);

/* R10 — ImitativeModel.forward mode search for model k (dim/model.py:76-141):
 * z_dev [B,64] (= _params), x0_dev [B,4,2] (the caller draws the base sample,
 * :100-104), goal_dev [B,G,2] or NULL.  One scalar loss (batch mean) and one
 * x_best for the whole batch (:124-137).  y_dev [B,4,2];
 * trace_loss_dev [num_steps] optional. */
int rip_dim_forward(rip_handle* h, int k, const float* z_dev, const float* goal_dev, const float* x0_dev, int B,
                    int G, int num_steps, float lr, float epsilon, float* y_dev, float* trace_loss_dev,
                    rip_stream_t stream);

/* R11 — the plan post-processing at the end of RIPAgent.__call__ / DIMAgent.__call__ (rip/agent.py:141-151,
 * dim/agent.py:74-84): scipy.interpolate.interp1d (linear) through the T = 4 waypoints at ticks 0, 10, 20, 30
 * (player_future_length 40 // T), sampled at ticks 0..29, z = 0 appended.  plan_dev [B,4,2] fp32 ->
 * out_dev [B,RIP_PLAN_ROWS,3] float64 — the dtype and, operation for operation, the arithmetic of the reference
 * (float32 knot difference, float64 slope / product / sum), so the result is bit-identical.  Stateless. */
#define RIP_PLAN_ROWS 30
int rip_interpolate_plans(const float* plan_dev, int B, double* out_dev, rip_stream_t stream);

/* Whole act() for B observations in one call: rip_encode_raw + rip_search (+ R11).
 * lidar_dev [B,H,W,C] (channels_last=1) or [B,C,H,W]; vec_dev [B,5];
 * goal_dev [B,G,2]; x0_dev [B,N,4,2]; plan_dev [B,4,2] (NULL to skip);
 * plan_interp_dev [B,RIP_PLAN_ROWS,3] float64 = what the agent's __call__ returns per observation (R11 fused into
 * the candidate selection; NULL to skip).  Scratch lives in the handle (B <= max_batch, N <= max_candidates). */
int rip_act(rip_handle* h, const float* lidar_dev, int channels_last, int H, int W, const float* vec_dev,
            const float* goal_dev, const float* x0_dev, int B, int N, int G, int algorithm, int num_steps, float lr,
            float epsilon, int enc_dtype, float* plan_dev, float* loss_best_dev, double* plan_interp_dev,
            rip_stream_t stream);

/* N3 — the DIM training step, oatomobile/baselines/torch/dim/train.py:175-213:
 *   z = model._params(...) in TRAIN mode (MobileNetV2 BatchNorm on batch statistics with the running-stat update,
 *   Dropout before the classifier), _, log_prob, logabsdet = decoder._inverse(y, z),
 *   loss = -mean(log_prob - logabsdet), loss.backward(), torch.optim.Adam(lr).step().
 * Parameters, gradients and the two Adam moments are caller-owned device vectors of rip_train_numel(C) floats in the
 * reference's state_dict order minus the int64 num_batches_tracked counters (the `packed_host` layout of
 * rip_load_model: conv weight, BN weight, BN bias, running_mean, running_var per conv, classifier, merger, GRUCell,
 * head), so that a data-parallel job all-reduces ONE gradient tensor between the two calls.  The trainer handle owns
 * the activation workspace for up to max_batch observations (fp32, 3 x 11.7 MB per observation).
 *
 * rip_train_forward_backward: visual_dev [B,C,100,100] (rip_transform output), vec_dev [B,5], y_dev [B,4,2] (the
 *   target, already perturbed by the caller: train.py:184-189), dropout_mask_dev [B,1280] keep/scale factors (0 or
 *   1/(1-p); NULL = no dropout).  batch_stats != 0: BatchNorm train mode, the running statistics inside params_dev are
 *   updated (momentum 0.1); 0: running statistics are used and left alone ("frozen" BatchNorm / evaluate_step).
 *   Writes grads_dev (running-statistic slots: 0), *loss_dev, z_dev [B,64] (optional).  grads_dev == NULL: forward
 *   only (evaluate_step, train.py:229-249): loss and z, no backward launches, no gradient buffer touched.
 *   Any B in [1, max_batch] (batch statistics of one observation are defined: the smallest map is 4 x 4).
 * rip_train_adam: torch.optim.Adam step `step` (1-based) on the entries with trainable_dev[i] != 0
 *   (rip_train_trainable_mask: everything but the running statistics); weight_decay is added to the gradient.
 * Both enqueue on `stream` without synchronising; rip_train_create / _destroy / _trainable_mask are setup calls. */
typedef struct rip_trainer rip_trainer;
size_t rip_train_numel(int in_channels);
int rip_train_create(rip_trainer** out, int in_channels, int max_batch, int device);
int rip_train_destroy(rip_trainer* t);
int rip_train_trainable_mask(const rip_trainer* t, unsigned char* mask_host, size_t numel);
int rip_train_forward_backward(rip_trainer* t, float* params_dev, float* grads_dev, const float* visual_dev,
                               const float* vec_dev, const float* y_dev, const float* dropout_mask_dev, int B,
                               int batch_stats, float* loss_dev, float* z_dev, rip_stream_t stream);
int rip_train_adam(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev,
                   const unsigned char* trainable_dev, size_t numel, int step, float lr, float beta1, float beta2,
                   float eps, float weight_decay, rip_stream_t stream);
/* Inspection: copies what the last rip_train_forward_backward (batch B) saved for conv layer `layer` (0 = features.0,
 * then the convs of features.1 .. features.18 in order; rip_train_num_layers of them) to dst_dev, NHWC [B,H,W,C]:
 * what = 0 the pre-BatchNorm conv output, 1 the post-activation output, 2 the gradient w.r.t. that output.
 * (A gradient through ReLU6 is only defined up to the kink decisions: two implementations whose forward values differ
 * in the 7th digit disagree on the mask of the ~1e-7 fraction of activations that sit on a kink, and each such element
 * moves a per-channel gradient by ~1/(B*H*W).  Tests read the masks back through this call to compare like with like.) */
int rip_train_peek(rip_trainer* t, int layer, int what, int B, float* dst_dev, size_t dst_numel, rip_stream_t stream);
int rip_train_num_layers(const rip_trainer* t);

/* Implementation knobs (results are identical within the parity tolerance; tests run every setting).
 *   RIP_OPT_SEARCH_KERNEL: 0 = auto (the split-f16 phase-sequential kernel when B*N >= 1280 and N % 16 == 0, else
 *     wave-per-chain),
 *     1 = wave-per-chain kernel (lowest latency, any K/N),
 *     (2 = round 1's fp32-MFMA wave-per-model pipeline: removed in round 5, RIP_EINVAL),
 *     3 = fp32-MFMA phase-sequential kernel (one wave per 16-candidate block runs all K models, operands in LDS, two
 *         waves per SIMD; N % 16 == 0, any K <= 8, trace outputs),
 *     4 = split-f16 phase-sequential kernel: the same decomposition with the GRU / head contractions on
 *         v_mfma_f32_16x16x32_f16, both operands carried as two binary16 terms (22 significant bits, fp32
 *         accumulation; same shapes and outputs as 3; round 6: the waypoint / bias terms as three-term operands, no fp32
 *         MFMA in the forward step).  auto picks it.
 *     5 = kernel 4 with the PAIRED workgroup shape forced (round 6, flow_pair.hip: a 16-candidate block on a pair of waves
 *         that split the hidden units, two waves per SIMD; same arithmetic per product, same gates; measured 1.27x slower
 *         than the one-wave shape, so auto never picks it).  All implement rip/agent.py:78-137.
 *   RIP_OPT_ENCODER_FUSED: how many leading MobileNetV2 inverted-residual blocks (0..17) run as ONE fused
 *     kernel each (expand -> LDS -> depthwise -> LDS -> project); the remaining, weight-dominated blocks run
 *     as one batched kernel per conv layer.  -1 (default) = auto.  fp32 encoder: 3 when B >= 8, else 0 — and,
 *     round 6, in the auto setting: stem + features.1 and features.2-7 as split-f16 row-streaming kernels at every
 *     launch size (small launches cut an observation into row bands), features.8-17 and features.18 + pool as split-f16
 *     tile / head kernels from 176 (model, observation) pairs per call — below that features.8-17 are an expansion +
 *     depthwise kernel and a layer-wise projection per block — (fp32 activations, pointwise convolutions as three
 *     binary16 MFMAs on two-term operands, depthwise / stem fp32: fp32-grade, z within 2e-5 of the fp32 oracle like the
 *     true-fp32 kernels), unless a model's pointwise weights reach 240 in magnitude or RIP_OPT_ENCODER_VARIANT bit 16 is
 *     set.  An explicit count (>= 0) runs exactly that split of the true-fp32 kernels of rounds 1-5.
 *     bf16 encoder: count >= 1 fuses the stem with features.1 (one kernel), blocks 1..6 of the count are the
 *     row-streaming kernel (features.2 .. features.7), blocks 7..15 the tile kernel (features.8 .. features.16);
 *     and 16 (features.17, round 5); features.18 runs as a GEMM with the pooled epilogue.  auto = everything, the tile kernel only when the call carries
 *     >= 96 (model, observation) pairs (an explicit count uses it regardless).
 *   RIP_OPT_SEARCH_REGROUP: retired in round 5 (rounds 3 / 4: regrouped the candidates of ONE workgroup by selected
 *     member between Adam steps; bit-identical results, fewer adjoints per block, no faster: the workgroup walks the
 *     model phases in lockstep).  Accepted, no effect.
 *   RIP_OPT_ENCODER_MEGA (experimental): the fp32 encoder of a small batch as ONE persistent launch instead of 55
 *     dependent ones (a dependent launch costs 4.4-4.8 us whatever it computes): ensemble member k runs on XCD k % 8
 *     only — its activations stay in that XCD's L2 and the barrier between two layers is a counter in that L2 (~1 us
 *     against 17-27 us for a device-wide one).  1 = every batch of up to 4 observations, 0 / -1 (default) = never:
 *     a layer between two such barriers still takes 2-7 us on the 32 CUs of one XCD (251-273 us per observation
 *     against 244 us for the launches), so the launches stay the default.  Same layer arithmetic as
 *     the layer-wise launches (tile shapes may differ: last-bit differences).  The kernel relies on
 *     the hardware placing workgroup i of a launch on XCD i % 8 (verified once per device at rip_create; the option is
 *     ignored where that does not hold); every workgroup re-checks its placement and every barrier wait is bounded
 *     (20 ms) — see rip_encoder_status.
 *   RIP_OPT_DEBUG_ENCODER_FAULT (tests only): value 1 / 2 raises the one-launch encoder's failure word as its kernel
 *     would after a placement miss / barrier timeout (RIP_ESTATE unless RIP_OPT_ENCODER_MEGA = 1 set the protocol up).
 *   RIP_OPT_ENCODER_VARIANT (development / tests; default 0 = what ships): bit mask of alternative bf16 encoder
 *     kernels kept for A/B runs — 2: stem + features.1 on round 3's front kernel,
 *     (1 and 4 selected round 1's row-streaming kernel for features.2-7 / 5-7: retired in round 6 — the matrix-core
 *     depthwise kernel is faster on all six blocks since the depthwise taps are bf16 values; accepted, no effect),
 *     8: features.17 as three layer-wise launches (round 4's persistent GEMMs + row-streaming depthwise) instead of a tile block.
 *     Same arithmetic definition; the teacher-forced block tests run every setting.
 *     16 (fp32 encoder): no split-f16 blocks — the true-fp32 kernels of rounds 1-5 at every launch size (A/B runs, taps
 *     of interior layers).
 *   RIP_OPT_KERNEL_LOG (tests; default 0): 1 = every rip_encode / rip_encode_raw* / rip_encode_tap* call records the
 *     encoder kernels it launches (name, template arguments, grid) — read with rip_kernel_log. */
enum { RIP_OPT_SEARCH_KERNEL = 0, RIP_OPT_ENCODER_FUSED = 1, RIP_OPT_SEARCH_REGROUP = 2, RIP_OPT_ENCODER_MEGA = 3,
       RIP_OPT_DEBUG_ENCODER_FAULT = 4, RIP_OPT_ENCODER_VARIANT = 5, RIP_OPT_KERNEL_LOG = 6 };
int rip_set_option(rip_handle* h, int option, int value);

/* The kernel-selection log of the handle's last encode / tap call under RIP_OPT_KERNEL_LOG = 1: one line per launch,
 * "kernel<template arguments> grid=(x,y,z) block=n".  Copies at most cap - 1 bytes + a terminating 0 into buf (buf may
 * be NULL) and returns the full length.  The tests assert with it that a parity case ran the kernels the headline
 * launch shape (B = 512, k_count = 4) selects. */
int rip_kernel_log(const rip_handle* h, char* buf, size_t cap);

/* 0, or non-zero once a one-launch encoder call (RIP_OPT_ENCODER_MEGA) found a workgroup off its XCD (1) or gave up
 * waiting at a layer barrier (2): the z of THAT call is invalid.  Read it after synchronising the stream of the call
 * (a host word, no device access); from then on the handle uses the layer-wise launches, so repeating the call gives
 * the valid result (oatomobile_amd/agents.py does exactly that).  ONE-SHOT: the failure is handed out once; every later
 * call returns 0 (the handle stays on the layer-wise launches).  A caller that never asks gets RIP_ESTATE from the
 * first entry point it calls after the failure became visible — once, so that the failure cannot pass unnoticed. */
int rip_encoder_status(rip_handle* h);

/* What rip_search would launch for B observations x N candidates under the handle's current options, and what that
 * launch executes on the matrix cores (bench.py's executed-flops count; rocprofv3 SQ_INSTS_MFMA is the check):
 * out[0] = kernel (1 wave-per-chain, 3 fp32-MFMA phase-sequential, 4 split-f16
 * phase-sequential); for kernels 3 / 4: out[1] = waves per workgroup, then (16x16x32 f16, 16x16x4 fp32) MFMA
 * instructions per 16-candidate block of out[2,3] one forward / inverse pass, out[4,5] the adjoint of an inverse pass,
 * out[6,7] the adjoint of F_0, out[8,9] the prefix step per (model, observation).  n_out >= 10. */
int rip_search_plan(const rip_handle* h, int B, int N, int32_t* out, int n_out);

/* Diagnostic (synchronises the device): the number of inverse-pass ADJOINTS the phase-sequential search kernels have
 * executed on this handle since the last reset — one per 16-candidate block, Adam step and ensemble member whose
 * adjoint some candidate of the block needed (rip/agent.py:121-129 back-propagates through the selected member only).
 * With rip_search_plan's per-pass instruction counts this is bench.py's executed-flops figure. */
int rip_search_stats(rip_handle* h, uint64_t* adjoint_passes, int reset);

/* Tracing hook: with RIP_ROCTX=1 in the environment rip_encode / rip_search / rip_train_* wrap their
 * launches in rocTX ranges (`rocprofv3 --marker-trace`); these two let the host layers (the collectives of
 * oatomobile_amd/distributed.py, replay batches) mark theirs through the same library.  They return 1 when tracing
 * is on (a range was opened / closed) and 0 when it is off (no-ops). */
int rip_trace_push(const char* name);
int rip_trace_pop(void);

/* Introspection used by bench.py / tests. */
int rip_num_models(const rip_handle* h);
int rip_in_channels(const rip_handle* h);
int rip_max_batch(const rip_handle* h);
int rip_max_candidates(const rip_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* RIP_HIP_H_ */
