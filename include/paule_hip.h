/*
 * paule_hip.h -- C-ABI of the MI355X (gfx950) planning engine: libpaule_hip.so
 *
 * The reference (quantling/paule) has NO native interface for this path: its planning
 * inner loop is ~60 lines of Python/torch nested inside Paule.plan_resynth()
 * (paule/paule.py:910-1211) and its only FFI is ctypes -> VocalTractLab
 * (paule/util.py:30-35), which is not on the path.  This header therefore DEFINES the
 * boundary a replacement has to provide; each entry point names the reference lines it
 * replaces.  Error convention = the one the reference already uses for its one C
 * library (nonzero int -> Python ValueError, paule/util.py:33-34, :235-236): every
 * function returns 0 on success and a nonzero code otherwise, with a message from
 * pl_last_error().  Nothing aborts; HIP errors are translated.
 *
 * All tensor arguments are DEVICE pointers (hipMalloc / torch.Tensor.data_ptr()),
 * dense row-major, float32 unless stated.  The caller owns everything it passes; the
 * library copies/repacks into its own layout, so the caller may free or mutate its
 * tensors after a call returns (needed because continued learning mutates pred_model
 * between outer iterations, paule/paule.py:1372-1377).  One handle <-> one device <->
 * one stream; calls on one handle are not re-entrant; different handles may be driven
 * from different threads / processes (one per GPU).  Handles of one process that share a
 * device may use different streams: the library chains the launches that run persistent
 * LSTM sweeps (they need their workgroups co-resident) behind each other per device.
 * Several PROCESSES on one GPU are not supported.  pl_step is stream-asynchronous;
 * pl_get_* and pl_step with output pointers enqueue their copies on the same stream.
 */
#ifndef PAULE_HIP_H
#define PAULE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PL_VERSION 100 /* 0.1.0 */

/* error codes */
enum {
    PL_OK = 0,
    PL_ERR_INVALID = 1,   /* bad argument / configuration */
    PL_ERR_HIP = 2,       /* HIP runtime error (message has hipGetErrorString) */
    PL_ERR_STATE = 3,     /* call sequence error (e.g. pl_step before weights/targets/cp are set) */
    PL_ERR_UNSUPPORTED = 4
};

/* arithmetic type of the LSTM / linear GEMMs and of the activation stash */
enum { PL_F32 = 0, PL_BF16 = 1 };

/* objective (paule/paule.py:600, :663, :719, :775-776) */
enum { PL_OBJ_ACOUSTIC = 0, PL_OBJ_ACOUSTIC_SEMVEC = 1, PL_OBJ_SEMVEC = 2 };

/* model ids for pl_set_lstm_weights / pl_set_linear */
enum { PL_MODEL_PRED = 0 /* ForwardModel, paule/models.py:326 */,
       PL_MODEL_EMBED = 1 /* EmbeddingModel, paule/models.py:413 */,
       PL_MODEL_INVERSE = 2 /* InverseModelMelTimeSmoothResidual, paule/models.py:177 (optional, pl_config.inv_layers) */,
       PL_MODEL_CP_TUBE = 3, PL_MODEL_TUBE_MEL = 4, PL_MODEL_TUBE_EMBED = 5 /* somatosensory feedback, pl_config.cp_tube_layers */ };

/* columns of one loss_log row (weighted sub-losses as logged at paule/paule.py:942-945, :988-992) */
enum { PL_LOSS_TOTAL = 0, PL_LOSS_MEL = 1, PL_LOSS_SEMVEC = 2, PL_LOSS_VEL = 3, PL_LOSS_JERK = 4,
       PL_LOSS_LOCAL_LINEAR = 5, PL_LOSS_SPEECH_CLASSIFIER = 6 /* 0 unless pl_set_speech_classifier */,
       /* with somatosensory feedback (pl_config.cp_tube_layers > 0; it excludes the speech classifier, paule/paule.py:117-118):
        * the tube path's weighted mel and semvec terms (paule/paule.py:624-644, weights :598-599) */
       PL_LOSS_TUBE_MEL = 6, PL_LOSS_TUBE_SEMVEC = 7,
       PL_LOSS_COLS = 8 };

typedef struct pl_handle pl_handle;

typedef struct pl_config {
    int32_t struct_size;      /* = sizeof(pl_config); guards against ABI drift */
    int32_t batch;            /* B independent utterances; reference: 1 (paule/paule.py:585-588) */
    int32_t n_frames;         /* T  CP frames  (xx_new.shape[1]) */
    int32_t cp_dim;           /* 30 (paule/util.py:50-52) */
    int32_t mel_dim;          /* 60 */
    int32_t sem_dim;          /* 300 */
    int32_t pred_layers;      /* ForwardModel num_lstm_layers (paule/models.py:338) */
    int32_t pred_hidden;      /* ForwardModel hidden_size */
    int32_t emb_layers;       /* EmbeddingModel num_lstm_layers; 0 = no embedder (acoustic objective only) */
    int32_t emb_hidden;
    int32_t dtype;            /* PL_F32 | PL_BF16 */
    int32_t objective;        /* PL_OBJ_* */
    float w_mel, w_sem, w_vel, w_jerk, w_ll; /* 5, 10, 80, 400, 1e5 (paule/paule.py:592-597) */
    float lr;                 /* learning_rate_planning, 0.01 (paule/paule.py:391) */
    float beta1, beta2, eps;  /* torch.optim.Adam defaults 0.9, 0.999, 1e-8 (paule/paule.py:797) */
    float clamp_lo, clamp_hi; /* -1.05, 1.05 (paule/paule.py:1202) */
    int32_t smiling;          /* paule/paule.py:1203-1208: ch 4 := -1, ch 1 := +1 after every step */
    int32_t device;           /* HIP device ordinal */
    int32_t use_graph;        /* 1: capture one inner iteration into a hipGraph and replay it; 0: eager launches */
    void *stream;             /* hipStream_t the work is enqueued on (NULL = default stream) */
    /* optional inverse model for the initialisation from the target acoustics (paule/paule.py:143-150,
     * :550-556): InverseModelMelTimeSmoothResidual(num_lstm_layers, hidden_size) with mel_smooth_filter_size = 3,
     * time_filter_size = 5 and Identity activations (the ctor defaults, paule/models.py:186-198).  inv_layers = 0: none. */
    int32_t inv_layers, inv_hidden;
    int32_t inv_mel_blocks;   /* mel_smooth_layers, 3 */
    int32_t inv_res_blocks;   /* resid_blocks, 5 (0 also drops resid_weighting, paule/models.py:206, :240) */
    /* embedder variants.  Both 0 = EmbeddingModel as Paule builds it (LSTM -> linear_mapping).
     * emb_post_size > 0: the head is post_linear(H -> post) -> LeakyReLU(0.01) -> output mapping(post -> sem_dim):
     *   EmbeddingModel(post_upsampling_size > 0) (paule/models.py:432-435, :443-446) and
     *   MelEmbeddingModelMelSmoothResidualUpsampling (paule/models.py:387-389, :405-407; post 8192).
     * emb_mel_blocks > 0: that many MelChannelConv1D(mel_dim, 3) blocks with residual connections in front of the LSTM
     *   (paule/models.py:384-385, :393-401; Identity activation). */
    int32_t emb_post_size;
    int32_t emb_mel_blocks;
    /* somatosensory feedback (paule/paule.py:227-273, :916-929): three more models beside the acoustic path,
     *   PL_MODEL_CP_TUBE    ForwardModel(cp_dim -> tube_dim, apply_half_sequence=False)  pred_tube      [B, T, tube_dim]
     *   PL_MODEL_TUBE_MEL   ForwardModel(tube_dim -> mel_dim, apply_half_sequence=True)  pred_tube_mel  [B, T/2, mel_dim]
     *   PL_MODEL_TUBE_EMBED EmbeddingModel(tube_dim -> sem_dim) on all T tube frames      pred_tube_semvec
     * and two more loss terms, w_mel * RMSE(pred_tube_mel, target_mel) + w_sem * RMSE(pred_tube_semvec, target_semvec)
     * (TUBE_MEL_WEIGHT = MEL_WEIGHT, TUBE_SEMANTIC_WEIGHT = SEMANTIC_WEIGHT, paule/paule.py:598-599, :631-642, :745-755), columns
     * 6 and 7 of the loss log.  Objectives acoustic_semvec and semvec only (the reference's `acoustic` criterion with
     * somatosensory feedback fails on an unassigned pred_tube_semvec, paule/paule.py:692).  cp_tube_layers = 0: off.
     * The tube embedder runs without dropout: the reference's default one has dropout 0.7 and is switched to .train() inside
     * the loop (paule/paule.py:266, :927), i.e. its loss is random there; a tube embedder with dropout 0 is deterministic. */
    int32_t tube_dim;                          /* 10 */
    int32_t cp_tube_layers, cp_tube_hidden;    /* 1, 360 */
    int32_t tube_mel_layers, tube_mel_hidden;  /* 1, 360 */
    int32_t tube_emb_layers, tube_emb_hidden;  /* 2, 720 */
} pl_config;

/* Fills *cfg with the reference's defaults (weights, lr, betas, clamp, dims 30/60/300). */
int pl_default_config(pl_config *cfg);

/* Allocates every device buffer of the planner (weights, stash, Adam state, scratch). */
int pl_create(const pl_config *cfg, pl_handle **out);
int pl_destroy(pl_handle *h);

/* torch.nn.LSTM layer `layer` of model `model_id`, torch layout: w_ih [4H, in], w_hh [4H, H],
 * b_ih [4H], b_hh [4H], gate order i,f,g,o (paule/models.py:344, :431).  Copied + repacked. */
int pl_set_lstm_weights(pl_handle *h, int model_id, int layer, const float *w_ih, const float *w_hh,
                        const float *b_ih, const float *b_hh);
/* post_linear [mel_dim, H] (paule/models.py:345) for PL_MODEL_PRED;
 * linear_mapping [sem_dim, H] (paule/models.py:437) for PL_MODEL_EMBED. */
int pl_set_linear(pl_handle *h, int model_id, const float *w, const float *b);

/* Optional speech-classifier term (use_speech_classifier=True; paule/models.py:887-910 LinearClassifier(60 -> 1),
 * paule/paule.py:604-622, :914-915): z = mean_t(w . mel_t + b), loss += weight * BCEWithLogits(z, 0) = weight * softplus(z).
 * w [mel_dim], b [1] device pointers (linear.weight / linear.bias); weight = 0.1 in the reference (paule/paule.py:596).
 * w = NULL switches the term off. */
int pl_set_speech_classifier(pl_handle *h, const float *w, const float *b, float weight);

/* target_mel [B, T/2, mel_dim]; target_semvec [B, sem_dim] or NULL (paule/paule.py:531-540). */
int pl_set_targets(pl_handle *h, const float *target_mel, const float *target_semvec);
/* cp [B, T, cp_dim]: xx_new (paule/paule.py:585-590).  Does not touch the optimiser state. */
int pl_set_cp(pl_handle *h, const float *cp);
/* past_cp [P, cp_dim] (per_utterance = 0, shared by all utterances) or [B, P, cp_dim]; P = 0 / NULL clears it
 * (paule/paule.py:575-583, :1210-1211). */
int pl_set_past_cp(pl_handle *h, const float *past_cp, int past_len, int per_utterance);
/* Adam m = v = 0, step count = 0 (a fresh torch.optim.Adam, paule/paule.py:797). */
int pl_reset_optimizer(pl_handle *h);

/* n_iters inner iterations (paule/paule.py:910-1211 without the log-step block):
 * forward, criterion, backward-data, Adam, clamp/smiling/past_cp.
 * loss_log  [n_iters, B, PL_LOSS_COLS] or NULL -- losses at the PRE-step CP of each iteration
 * grad_out  [B, T, cp_dim] or NULL            -- xx_new.grad of the LAST iteration (paule/paule.py:1056-1063) */
int pl_step(pl_handle *h, int n_iters, float *loss_log, float *grad_out);

/* Waits for everything enqueued on the handle's stream and reports device-side failures that cannot be returned
 * asynchronously (a bounded in-kernel wait of a persistent LSTM sweep timed out -> PL_ERR_HIP).
 *
 * ONE PROCESS PER GPU.  The LSTM sweeps are persistent launches whose workgroups wait for each other inside the launch
 * (one workgroup per CU; the fused acoustic launches take up to all 256 CUs).  They finish only when all their workgroups
 * are resident at the same time.  Handles of ONE process are safe: every entry point that launches sweeps chains behind the
 * previous one on the same device, whatever stream it runs on.  A SECOND process on the same GPU can hold CUs a launch is
 * waiting for.  The role-fused launches notice that at their top -- every workgroup signs in and waits at most
 * PAULE_HIP_CENSUS_MS (50 ms) for the others -- and this call returns PL_ERR_STATE; the per-layer sweeps run into the
 * bounded wait (PAULE_HIP_SPIN_MS, default 2 s) and this call returns PL_ERR_HIP.  Either way the
 * plan of that pl_step is not valid, call pl_set_cp / pl_reset_optimizer and step again once the GPU is yours.  Host code
 * must call this (or pl_get_cp, which a caller follows with it) before it trusts results: the Python layers do
 * (Paule.plan_resynth, plan_sharded). */
int pl_synchronize(pl_handle *h);

int pl_get_cp(pl_handle *h, float *cp_out);
/* Forward only at the current CP (paule/paule.py:822-824, :1460-1464): pred_mel [B, T/2, mel_dim],
 * pred_semvec [B, sem_dim] (may be NULL; requires an embedder). */
int pl_get_pred(pl_handle *h, float *pred_mel_out, float *pred_semvec_out);
/* The predictive model without the half sequence: post_linear(lstm(cp)) of every frame, frames_out [batch, n_frames, mel_dim]
 * (device).  Replaces ForwardModel(apply_half_sequence=False).forward, paule/models.py:348-356 -- the form the reference
 * builds its cp -> tube model in (paule/paule.py:232-237). */
int pl_get_pred_frames(pl_handle *h, float *frames_out);
/* The same for the somatosensory path (paule/paule.py:916-929): pred_tube [B, T, tube_dim], pred_tube_mel [B, T/2, mel_dim],
 * pred_tube_semvec [B, sem_dim]; any of them may be NULL. */
int pl_get_tube_pred(pl_handle *h, float *pred_tube_out, float *pred_tube_mel_out, float *pred_tube_semvec_out);
/* tube_mel_model(tube) and tube_embedder(tube, T) on an arbitrary tube [B, T, tube_dim], e.g. the one extracted from the synthesis
 * (prod_tube_mel, prod_tube_semvec, paule/paule.py:1084, :1147-1150); either output may be NULL. */
int pl_embed_tube(pl_handle *h, const float *tube, float *tube_mel_out, float *tube_semvec_out);
/* EmbeddingModel.forward on an arbitrary mel [B, T/2, mel_dim] (paule/paule.py:533-535, :1131):
 * lens [B] int32 device pointer or NULL (= T/2 for every utterance, paule/paule.py:922-924). */
int pl_embed_mel(pl_handle *h, const float *mel, const int32_t *lens, float *semvec_out);

/* ---- embedder variants (pl_config.emb_post_size / emb_mel_blocks) --------------------------------------
 * With emb_post_size > 0, pl_set_linear(PL_MODEL_EMBED) takes post_linear [post, H] (the linear that reads the LSTM output) and
 * pl_set_embedder_output the mapping behind the LeakyReLU: linear_mapping.weight [sem_dim, post] of
 * EmbeddingModel(post_upsampling_size > 0) / upsampling.weight of MelEmbeddingModelMelSmoothResidualUpsampling, bias [sem_dim].
 * pl_set_embedder_conv: MelBlocks[block].ConvLayers[idx], w [mel_dim/3, 3, 5], b [mel_dim/3] (paule/models.py:148-150).
 * The planning loop differentiates through all of it (head, LSTM stack, mel blocks) exactly as through the plain embedder. */
int pl_set_embedder_output(pl_handle *h, const float *w, const float *b);
int pl_set_embedder_conv(pl_handle *h, int block, int idx, const float *w, const float *b);

/* ---- inverse model: initial CP from the target mel -------------------------------------------------
 * The LSTM stack and post_linear are uploaded with pl_set_lstm_weights / pl_set_linear and model_id PL_MODEL_INVERSE
 * (lstm.weight_ih_l{k} [4H, 3 * mel_dim], ..., post_linear [cp_dim, H]).  Convolutions, torch Conv1d layout, device ptrs:
 *   PL_CONV_MEL   MelBlocks[block].ConvLayers[idx]             w [mel_dim/3, 3, 5], b [mel_dim/3]  (paule/models.py:148-150)
 *   PL_CONV_RES   ResidualConvBlocks[block].band_conv1d_{idx+1} w [cp_dim, 1, 5],    b [cp_dim]     (paule/models.py:123-124)
 *   PL_CONV_RW    resid_weighting (block = idx = 0)             w [cp_dim, 2, 5],    b [cp_dim]     (paule/models.py:207-208) */
enum { PL_CONV_MEL = 0, PL_CONV_RES = 1, PL_CONV_RW = 2 };
int pl_set_inverse_conv(pl_handle *h, int kind, int block, int idx, const float *w, const float *b);
/* initial_cp = inv_model(target_mel) (paule/paule.py:552-553): mel [B, n_mel_frames, mel_dim] -> cp_out [B, 2 * n_mel_frames,
 * cp_dim]; n_mel_frames <= T/2.  clip != 0 applies .clip(min=-1, max=1) of paule/paule.py:555. */
int pl_inverse_forward(pl_handle *h, const float *mel, int n_mel_frames, float *cp_out, int clip);

/* ---- continued learning of the predictive model (paule/paule.py:1353-1379) ---------------------
 * One optimiser step of `pred_model` on a mini-batch, entirely on the device, replacing
 *     Y_hat = self.pred_model(batch_input, lens_input_j); self.pred_optimizer.zero_grad()
 *     pred_loss = self.pred_criterion(Y_hat, batch_output); pred_loss.backward(); self.pred_optimizer.step()
 * (paule/paule.py:1372-1377) with pred_criterion = RMSELoss(eps=0) over the whole batch (:288) and
 * pred_optimizer = torch.optim.Adam (:287; lr = learning_rate_learning, :473-474).
 *   n_rows      samples of the mini-batch, 1 <= n_rows <= batch rounded up to a multiple of 16 (the handle's padded rows: a
 *               planner built for the reference's ONE utterance still trains on its mini-batches of 8, paule/paule.py:404)
 *   n_frames    their (padded) length, 2 <= n_frames <= T: pad_batch_online pads a batch to its longest sample (:1362-1368;
 *               same_size_batching keeps that small, :1356-1358); padded frames take part in the loss as in the reference
 *   cp          [n_rows, n_frames, cp_dim]     batch_input  (cp_norm of produced samples)
 *   mel_target  [n_rows, n_frames/2, mel_dim]  batch_output (melspec_norm_synthesized)
 *   loss_out    device float or NULL: pred_loss of this step (before the update)
 * The handle keeps f64 masters of every parameter and the Adam moments; the packed compute copies the planner reads
 * are refreshed by the same launch sequence, so the next pl_step / pl_get_pred uses the new weights (no re-upload,
 * contrast paule/paule.py:1372-1377 mutating pred_model in place).  Stream-asynchronous.  Clobbers only scratch. */
int pl_train_pred_step(pl_handle *h, int n_rows, int n_frames, const float *cp, const float *mel_target, float lr,
                       float beta1, float beta2, float eps, float *loss_out);
/* Fresh torch.optim.Adam state for the parameters (moments = 0, step count = 0). */
int pl_reset_pred_optimizer(pl_handle *h);
/* Adam state of that optimiser, for saving / restoring / carrying it from one handle to the next (the reference's
 * pred_optimizer lives as long as the Paule instance, paule/paule.py:284-287; docs/examples/minimal_example.py:51 saves it):
 * which = 1 exp_avg, 2 exp_avg_sq; layer >= 0: the four LSTM tensors of that layer, layer = -1: post_linear (w_ih = weight,
 * b_ih = bias, the other two ignored).  float32 device pointers, torch layout.  The step count is torch's state["step"]. */
int pl_get_pred_optimizer_state(pl_handle *h, int layer, int which, float *w_ih, float *w_hh, float *b_ih, float *b_hh);
int pl_set_pred_optimizer_state(pl_handle *h, int layer, int which, const float *w_ih, const float *w_hh, const float *b_ih,
                                const float *b_hh);
int64_t pl_get_pred_optimizer_step(const pl_handle *h);
int pl_set_pred_optimizer_step(pl_handle *h, int64_t step);
/* The same step for any ForwardModel of the handle: model_id PL_MODEL_PRED (= pl_train_pred_step), PL_MODEL_CP_TUBE
 * (input cp [n_rows, n_frames, cp_dim], target tube [n_rows, n_frames, tube_dim]: no half sequence) or PL_MODEL_TUBE_MEL
 * (input tube [n_rows, n_frames, tube_dim], target mel [n_rows, n_frames/2, mel_dim]) -- continue_learning_tube,
 * paule/paule.py:1381-1404 with tube_optimizer / tube_mel_optimizer = Adam(lr 0.001) and RMSELoss (:296-306) -- and the
 * optimizer state of each (reset, exp_avg / exp_avg_sq transfer, step count), as for the predictive model. */
int pl_train_model_step(pl_handle *h, int model_id, int n_rows, int n_frames, const float *input, const float *target, float lr,
                        float beta1, float beta2, float eps, float *loss_out);
int pl_reset_model_optimizer(pl_handle *h, int model_id);
int pl_get_model_optimizer_state(pl_handle *h, int model_id, int layer, int which, float *w_ih, float *w_hh, float *b_ih, float *b_hh);
int pl_set_model_optimizer_state(pl_handle *h, int model_id, int layer, int which, const float *w_ih, const float *w_hh,
                                 const float *b_ih, const float *b_hh);
int64_t pl_get_model_optimizer_step(pl_handle *h, int model_id);
int pl_set_model_optimizer_step(pl_handle *h, int model_id, int64_t step);
/* Current parameters in torch layout (the layouts of pl_set_lstm_weights / pl_set_linear), float32 device pointers:
 * how the host's torch module is brought back in sync after continued learning. */
int pl_get_lstm_weights(pl_handle *h, int model_id, int layer, float *w_ih, float *w_hh, float *b_ih, float *b_hh);
int pl_get_linear(pl_handle *h, int model_id, float *w, float *b);

/* Test / debugging aid: copies a named internal buffer, converted to float32, into out (device pointer).
 * Returns the element count through *n_out; out may be NULL to query the size only. */
int pl_debug_read(pl_handle *h, const char *name, float *out, int64_t max_elems, int64_t *n_out);

/* Measurement aid for bench.py's roofline line: launches ONE kernel of the hot path `reps` times back to back
 * on the handle's stream with real operands (cycling over the time steps of layer 0 of `model_id`), bracketed
 * by hipEvents on that stream, and returns the average launch-to-launch time (ms, inter-launch gap included)
 * and the algorithmic FLOPs of one launch (2 * B * 4H * H).  Clobbers only scratch that every pl_step rebuilds. */
enum { PL_KERNEL_LSTM_FWD_STEP = 0, PL_KERNEL_LSTM_BWD_STEP = 1,
       /* persistent sweeps: one launch = all T steps of layer 0 (+ its ~5 us counter-zeroing launch); FLOPs = 2*B*4H*H*(T-1) */
       PL_KERNEL_LSTM_FWD_SWEEP = 2, PL_KERNEL_LSTM_BWD_SWEEP = 3,
       /* the fused acoustic forward launch (predictor + mel head + embedder layers as roles of one grid; bf16; batches of up to
        * 48 rows on 16-row tiles, 49+ rows on 32-row tiles): one launch = all steps of all its layers (+ the flag-zeroing launch); FLOPs = 2 * B * sum over layers of
        * 4H (in + H) T_layer + 2 * B * H * mel_dim * T; PL_ERR_UNSUPPORTED when the handle does not use it; model_id ignored */
       PL_KERNEL_FUSED_FWD = 4,
       /* the fused acoustic backward launch (embedder recurrences top-down, their dL/dh products, backward mel head, predictor
        * recurrence; bf16; by default batches of up to 128 rows -- up to 48 on 16-row tiles -- and, for a stacked predictor of another
        * width than the embedder, every batch its roles fit the chip): FLOPs = 2 * B * (sum over layers of 4H * H * (T_layer - 1)
        * + the products between the roles); PL_ERR_UNSUPPORTED when the handle does not use it; model_id ignored */
       PL_KERNEL_FUSED_BWD = 5 };
int pl_bench_kernel(pl_handle *h, int kernel, int model_id, int reps, float *avg_ms_out /* host */,
                    double *flops_per_launch_out /* host */);

/* Which launch schedule the handle planned for its shapes (host array of n >= 0 ints; entries beyond PL_PLAN_COUNT read 0).
 * Test / measurement aid: the planner declines the role-fused launches silently on many conditions (CU budget, model
 * widths, batch below the crossover ...), and a parity test that compares "fused" with "per layer" must be able to tell
 * that the fused launch really ran.  No reference counterpart (the reference has one schedule: torch's). */
enum { PL_PLAN_FUSED_FWD = 0,      /* 1: role-fused forward launch, 0: per-layer forward sweeps / pipelines */
       PL_PLAN_FUSED_BWD = 1,      /* 1: role-fused backward launch */
       PL_PLAN_FWD_CHAINS_PRED = 2, PL_PLAN_FWD_CHAINS_EMB = 3,   /* batch groups per workgroup of the forward roles */
       PL_PLAN_BWD_CHAINS_PRED = 4, PL_PLAN_BWD_CHAINS_EMB = 5,   /* ... of the backward roles (their own counts since round 4) */
       PL_PLAN_FWD_WORKGROUPS = 6, PL_PLAN_BWD_WORKGROUPS = 7,    /* role-bearing workgroups of the fused launches */
       PL_PLAN_BWD_WAVES = 8,      /* waves per workgroup of the per-layer reduce-scatter backward sweep (4 or 8) */
       PL_PLAN_N_CU = 9,
       PL_PLAN_RETAINED_EXECS = 10, /* process-wide: graph execs with parallel branches that retired handles left allocated (host
                                    * memory; the HIP runtime crashes at a later branched launch once such execs are destroyed) */
       PL_PLAN_FUSED_ROWS = 11,     /* batch rows of the fused launches' LSTM roles: 32, 16 (batches of up to 16 rows), 0 = no fused launch */
       PL_PLAN_FWD_PER_CU = 12,     /* workgroups per CU the fused forward launch is written for: 1, 2 (lstm_fused2.hip) or 0 = no fused forward launch */
       PL_PLAN_BWD_PREFETCHERS = 13, /* prefetcher workgroups beside the predictor's streamed per-layer backward sweep (round 5: they warm the XCD's L2 with
                                     * the stash rows ahead of the cell waves; speed only), 0 = none / another kernel carries the backward pass */
       PL_PLAN_COUNT = 14 };
int pl_plan_info(const pl_handle *h, int32_t *out /* host */, int n);

/* Bytes of device memory held by the handle. */
int64_t pl_device_bytes(const pl_handle *h);
/* Algorithmic FLOPs (forward + backward-data GEMMs, 2 flops / MAC) of one inner iteration at this handle's shapes. */
double pl_flops_per_iteration(const pl_handle *h);

const char *pl_last_error(void); /* thread-local */
int pl_version(void);            /* PL_VERSION; answered by the loader itself: touches no HIP runtime */
/* HIP_VERSION (major * 10000000 + minor * 100000 + patch) of the toolchain the kernels were compiled with.  The loader compares its major
 * with the major of the HIP runtime it binds the kernels to and refuses a mismatch with a clear pl_last_error() instead of undefined
 * behaviour inside the runtime (a host may bring a runtime of another ROCm release than /opt/rocm's, e.g. a PyTorch wheel's). */
int pl_hip_version_built(void);

#ifdef __cplusplus
}
#endif
#endif /* PAULE_HIP_H */
