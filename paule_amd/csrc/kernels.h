// Host-side launch prototypes of every device kernel of the planner (implemented in *.hip).
// All launches are asynchronous on `stream`; none allocates or synchronises (graph-capture safe).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pl {

enum DType { F32 = 0, BF16 = 1 };
inline size_t dtype_size(int dt) { return dt == BF16 ? 2 : 4; }

// ---- gemm.hip -------------------------------------------------------------------------------
// C[M,N] = A[M,K] * W[N,K]^T (+ bias[N]);  A, W of activation type `dt`; C float if out_f32 else `dt`.
// lda/ldw/ldc in elements; K % 32 == 0; rows 16-byte aligned.
// gemm_big.hip: the large bf16 products on 256 x 256 tiles (bit-identical to gemm.hip's tiles); launch_gemm_nt hands them over by itself
bool gemm_big_takes(int M, int N, int K);
void gemm_big_init();   // once per process, outside any stream capture (function attribute: 128 KB of LDS)
void gemm_set_big(bool on);
void launch_gemm_nt_big(hipStream_t stream, bool out_f32, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                        int M, int N, int K);
void launch_gemm_nt(hipStream_t stream, int dt, bool out_f32, const void* A, int lda, const void* W, int ldw,
                    const float* bias, void* C, int ldc, int M, int N, int K);

// ---- lstm.hip -------------------------------------------------------------------------------
struct LstmStepArgs {
    int Bp, Hp;            // padded batch rows in a time slab, padded hidden size
    // forward: G_t holds W_ih x_t + b on entry and the activated gates (i,f,g,o) on exit   [Bp][4*Hp]
    // backward: G_t holds the activated gates on entry and dA_t on exit; G_next = dA_{t+1} (or null at t = T-1)
    void* G_t;
    const void* G_next;
    const void* W;         // forward: Whh packed [4*Hp][Hp]; backward: Whh^T packed [Hp][4*Hp]
    const void* h_prev;    // forward: h_{t-1} [Bp][Hp] (null at t = 0)
    void* h_out;           // forward: h_t [Bp][Hp]
    const float* c_in;     // forward: running c_{t-1} f32 [Bp][Hp] (null at t = 0); backward: running dc_{t+1}*f_{t+1}
    float* c_out;          // forward: running c_t; backward: running dc_t * f_t
    void* c_stash_t;       // forward (write) / backward (read): c_t in activation type [Bp][Hp]
    const void* c_stash_prev;  // backward: c_{t-1} stash (null at t = 0)
    const void* dh_ext;    // backward: dL/dh_t from the layer above [Bp][Hp] activation type (null = zeros)
    unsigned long long* stamps;  // diagnostic builds (-DPL_STAMPS) only: [block][8] s_memrealtime stamps; null otherwise
};
void launch_lstm_fwd_step(hipStream_t stream, int dt, const LstmStepArgs& a);
void launch_lstm_bwd_step(hipStream_t stream, int dt, const LstmStepArgs& a);

// ---- lstm_persist.hip -----------------------------------------------------------------------
// One launch = all T steps of one layer (bf16, register-resident W_hh, in-launch exchange).
// Arguments of the CP update (total gradient + Adam + projection, elementwise.hip, cp_update.h)
struct AdamArgs {
    int B, T, C, Bp, Cp;
    double lr, beta1, beta2, eps, clamp_lo, clamp_hi;
    float w_vel, w_jerk, w_ll;
    int smiling;
    const float* dX;       // model gradient f32 time-major [T][Bp][Cp]
    const float* dX2;      // second model gradient (the CP -> tube model of the somatosensory path), same layout, or null
    double* x;             // CP master [B][T][C]
    double* m;
    double* v;
    double* grad;          // total gradient [B][T][C] (model + smoothness)
    const double* dwork;   // correlations written by the loss reduction of this iteration
    int* step_count;       // device counter k (incremented by the update kernel)
    int* iter_slot;        // device counter (incremented by the update kernel)
    const double* past;    // past_cp [B or 1][P][C] or null
    int past_len, past_per_utt;
};

struct LstmSweepArgs {
    int Bp, T;
    int group_rows;        // batch rows per group (8..32, multiple of 8): lstm_sweep_group_rows()
    void* G;               // [T][Bp][4*Hp]: forward: W_ih x + b in, activated gates out; backward: gates in, dA out
    const void* W;         // forward: Whh [4*Hp][Hp]; backward: Whh^T [Hp][4*Hp]
    void* h;               // forward: h stash [T][Bp][Hp] (written; h_{t-1} is read back by the whole group)
    void* c;               // c stash [T][Bp][Hp] (forward writes, backward reads)
    const void* dh_ext;    // backward: dL/dh from above [T][Bp][Hp], or null
    const void* dh_last;   // backward: [Bp][Hp] applied at t = T-1 only (when dh_ext is null), or null
    int* counters;         // arrival flags [groups][T][flag_stride], zeroed before the launch
    int flag_stride;       // >= Hp / 32 (workgroups per group), multiple of 16
    int* xcc_tab;          // [groups][64] XCD ids (+1) of the group's workgroups, zeroed with the counters
    int xcd_fast;          // 1: groups that find themselves on one XCD hand off with plain stores (speed only)
    int* status;           // set to 1 when a bounded spin timed out (the sweep is then abandoned)
    unsigned long long spin_ticks;   // bound of every in-kernel wait, in 100 MHz s_memrealtime ticks
    unsigned poll_mask;              // the abort / timeout check runs on polls with (n & poll_mask) == 0
    unsigned long long* stamps;      // diagnostic builds (-DPL_STAMPS) only: [block][8] accumulated phase ticks
    void* xchg;            // reduce-scatter backward only: partial-tile exchange [2 slots][groups][P][32 rows][Hp]
    // forward, fused input projection (narrow inputs, in_p = 32 / 64): null x_in = G holds the precomputed projection
    const void* x_in;      // layer input, time-major [T][Bp][in_p]
    const void* Wih;       // packed [4*Hp][in_p]
    const float* bias;     // b_ih + b_hh, [4*Hp]
    int in_p;
    // time chunk (16-row bf16 and f32 kernels only): the launch runs steps t0 .. t1-1 of the T-step recurrence (t1 = 0: all
    // T steps).  The flags / stashes of the neighbouring chunk are in place when the launch starts (stream order); the f32 cell
    // state (forward: c, backward: dL/dc) crosses the chunk border through `carry` [Bp][Hp] f32.
    int t0, t1;
    float* carry;
    // f32 kernels: 1 .. 4 = that many batch rows are in use and the batch is one group: the recurrent products run as
    // f32 FMAs on those rows only (the 16x16x4 MFMA costs the same for 1 row as for 16 and bounds the step); 0 = MFMA
    int n_valid;
    int stash_via_lds;     // forward, 32-row kernel: 1 = the five stash arrays leave through LDS as 64-byte row pieces
    int chains;            // lstm_chain_f32.hip: batch groups a workgroup serves in turn with one copy of its weights
    int bwd_waves;         // lstm_persist_rs.hip: waves per workgroup, 4 or 8 (0 = 8, the default; PAULE_HIP_BWD_WAVES)
    int* tflags;           // lstm_persist_rs.hip, streamed form: per-tile flags [2 slots][groups][P destinations][32] (zeroed with the
                           // counters); null = the whole-workgroup hand-off (one flag per workgroup and step)
    // lstm_persist_rs.hip, streamed form, first layer of a model with a narrow input (round 4): the workgroup's partial input gradient
    // dA_t[its 128 gate rows] * W_ih rides along as one more tile of the step (the free tile slot of the last wave: 23 tiles on 8 x 3
    // slots) and leaves as f32 to xpart [T][groups][P][32 x 32]; launch_dx_reduce then sums the P partials in a fixed order.  The batched
    // product dA * W_ih with its re-read of the whole dA stash (452 MB at cfg3) disappears.  null = off
    const void* WihT;      // [in_p = 32][4*Hp] packed like Whh^T
    float* xpart;
    int skip_dA;           // with xpart: 1 = dA_t is NOT written to the stash (nothing reads the predictor's dA in a planning iteration)
    // lstm_persist_rs.hip, streamed form (round 5): n_pf extra workgroups behind the sweep's own are PREFETCHERS -- they follow the groups' tile
    // flags and read the stash rows (gates, c, dL/dh from above) of step t - pf_dist of their group, one dword per 128-byte line, so that the
    // lines sit in the XCD's L2 when the cell waves load them: a cell wave's loads return in order, and its first flag poll of a step queues
    // behind the step's stash loads -- from HBM ~2 us, from L2 ~0.3.  Speed only: no result depends on a prefetcher.  0 = none
    int n_pf, pf_dist;
    int token_handoff;     // lstm_persist_rs.hip, token form (round 4): 1 = the tiles carry their own step token, no flags, no drains; xchg
                           // is then a buffer ONLY this form uses (zero at the start of every launch: the kernel leaves it retired)
};
bool lstm_sweep_supported(int dt, int Hp);
// workgroups to launch (multiple of Hp / 32, all co-resident on n_cu CUs); 0 = does not fit
int lstm_sweep_grid(int Hp, int Bp, int n_cu, bool spread_small = false);
int lstm_sweep_group_rows(int Hp, int Bp, int n_cu);
void launch_lstm_sweep(hipStream_t stream, bool backward, int Hp, int grid, const LstmSweepArgs& a);
// backward sweep in reduce-scatter form (lstm_persist_rs.hip): same arguments + a.xchg of lstm_rs_exchange_bytes()
size_t lstm_rs_exchange_bytes(int Hp, int Bp);
void launch_lstm_bwd_rs_sweep(hipStream_t stream, int Hp, int grid, const LstmSweepArgs& a);
// the ride-along input gradient (LstmSweepArgs::xpart): supported shape, scratch size, and the fixed-order sum of the partials into
// dX f32 [T][Bp][32] (what the batched product wrote)
bool lstm_rs_ride_along_supported(int Hp, int in_p);
size_t lstm_rs_xpart_bytes(int Hp, int Bp, int T);
void launch_dx_reduce(hipStream_t stream, const float* xpart, int Hp, int Bp, int T, float* dX);
// bf16 sweeps on groups of 16 rows for batches of up to 128 rows (lstm_persist16.hip); same arguments, group_rows <= 16
bool lstm_sweep16_wanted(int Hp, int Bp, int n_cu);
int lstm_sweep16_grid(int Hp, int Bp, int n_cu, bool spread_small);
void launch_lstm_sweep16(hipStream_t stream, bool backward, int Hp, int grid, const LstmSweepArgs& a);
// f32 sweeps (lstm_persist_f32.hip): groups of 16 rows, Hp / 16 workgroups per group, backward in reduce-scatter form
bool lstm_sweep_f32_supported(int Hp);
int lstm_sweep_f32_grid(int Hp, int Bp, int n_cu);
size_t lstm_f32_exchange_bytes(int Hp, int Bp);
void launch_lstm_sweep_f32(hipStream_t stream, bool backward, int Hp, int grid, const LstmSweepArgs& a);
// f32 chains (lstm_chain_f32.hip): more 16-row groups than the chip holds at once.  plan: chains per workgroup (0 = not this path)
bool lstm_chain_f32_supported(int Hp);
int lstm_chain_f32_plan(int Hp, int Bp, int n_cu, int forced_chains, int* grid);
void launch_lstm_chain_f32(hipStream_t stream, bool backward, int Hp, int grid, const LstmSweepArgs& a);
// ---- lstm_fused.hip ------------------------------------------------------------------------
// The acoustic path's LSTM sweeps of one direction as ONE persistent launch (bf16, all models of one hidden size): the
// workgroups of the grid take ROLES (a layer's recurrence, a layer's input projection, the mel head with its pooling, ...)
// from a host-built table, every role runs all its time steps, and the roles hand over per time step through arrival flags
// -- predictor step t, mel head of frame t / 2, embedder layer 1, projection for layer 2, layer 2 overlap instead of running
// one sweep after the other.  A workgroup serves C batch groups ("chains") with ONE copy of its weights in registers: while
// one chain waits for its exchange the others compute.
constexpr int kFusedMaxRoles = 12;
constexpr int kFusedMaxChains = 8;
constexpr int kFusedRing = 4;   // slots of the cross-role partial-tile exchanges (a producer runs at most this far ahead)
enum { FR_NONE = -1, FR_LSTM_FWD = 0, FR_PROJ_FWD = 1, FR_HEAD_FWD = 2, FR_LSTM_BWD = 3, FR_DX_BWD = 4, FR_HEAD_BWD = 5 };
// One set of arrival flags a chain-step (group g, step t) of a role waits for: flags[(g * T + tt) * flag_stride + i] with
// tt = (t >> t_shr) + t_add, for i < n -- or the single flag i = the workgroup's own slice p when per_p is set.  The entry
// is skipped when tt is outside [0, T) (no step before the first, ring not yet wrapped, ...) or flags is null.
struct FusedWait {
    const int* flags;
    int T, n, per_p, t_shr, t_add;
};
struct FusedRole {
    int type;              // FR_*
    int ksx;               // FR_LSTM_FWD: k-steps (of 16) of the fused narrow input projection (2 / 4), 0 = G holds the projection
    int wide;              // forward launch with two hidden sizes: 0 the role works at the predictor's width, 1 at the embedder's
    int C;                 // chains (batch groups of 32 rows) per workgroup; set s serves groups s * C .. s * C + C - 1
    int T;                 // time steps of the role
    int* flags;            // the role's own arrival flags [groups][T][flag_stride] (zeroed before the launch)
    int* flags2;           // FR_DX_BWD: second set, raised when the role's REDUCED dL/dh rows of a step are in place
    FusedWait wait[3];     // [0]: <= 32 flags, [1]: <= 31 flags, [2]: one flag
    int src_sc1;           // LSTM roles: x / G / dh_ext rows are produced by a role of this launch: load them write-through (sc1)
    // LSTM roles
    void* G;               // [T][Bp][4 Hp]
    const void* W;         // forward Whh [4 Hp][Hp], backward Whh^T [Hp][4 Hp]
    void* h;               // [T][Bp][Hp]
    void* c;               // [T][Bp][Hp]
    const void* x_in;      // forward, ksx > 0: [T][Bp][16 ksx]
    const void* Wih;       // forward, ksx > 0: [4 Hp][16 ksx]
    const float* bias;     // [4 Hp] (LSTM with fused input, projection role) / [out_p] (head)
    // projection / head roles: out = src_h[t] * Wg^T + bias
    const void* src_h;     // [T][Bp][Hp] of the producing layer
    const void* Wg;        // projection: Wih [4 Hp][Hp] of the consuming layer; head: Wlin [out_p][Hp]; backward head: Wlin^T [Hp][out_p];
                           // FR_DX_BWD: Wih^T [Hp][4 Hp] of the layer above; FR_LSTM_BWD with dmel_out: Wih^T [out_p][4 Hp] of this layer
    void* out;             // projection: G of the consuming layer; head: pooled mel, time-major activation [T / 2][Bp][out_p];
                           // backward head: dL/dh of the predictor's top layer per POOLED frame [T][Bp][Hp]
    float* out_bm;         // head: pooled mel f32 batch-major [B][T / 2][out_dim]
    int out_dim, out_p;
    // backward roles
    const void* dh_ext;    // LSTM: dL/dh from above [T or T / 2][Bp][Hp]; backward head: loss part of dL/dY, f32 [2 T][Bp][out_p] (row 2 t read)
    int dh_ext_half;       // 1: step t of the recurrence reads dh_ext row t / 2
    int dh_ext_rows;       // rows of dh_ext (steps beyond read zeros)
    const void* dh_last;   // [Bp][Hp], applied at t = T - 1
    int dA_sc1;            // LSTM: the dA stash is read by a role of this launch: store it write-through and drain it before the flag
    void* xchg;            // LSTM: reduce-scatter exchange of the recurrence [2][groups][P][P][32][32]
    void* xchg_ext;        // LSTM: partial tiles of dL/dh from the layer above's FR_DX_BWD role [ring][groups][P][P][32][32] (null: none);
                           // FR_DX_BWD: where it writes them
    void* xchg_mel;        // LSTM: partial tiles of this layer's input gradient, out_p / 32 tiles per source [ring][groups][out_p / 32][P][32][32]
                           // (null: none); backward head: where it reads them
    // 16-row LSTM roles (launch_fused_fwd16 / _bwd16: batches of up to 16 rows): the verified same-XCD form of the role's OWN exchange
    int* fast_flags;       // second flag set [T][flag_stride], stored plain / polled nt once the role found itself on one XCD
    int* xtab;             // [P] XCD ids (+ 1) of the role's workgroups, written with their first hand-off
    void* hx;              // forward: private copy of the h hand-off [2 slots][16][Hp] (plain stores, nt loads); `h` stays write-through
};
struct FusedArgs {
    int Bp, B, n_groups, flag_stride, n_roles, grid;
    int gpp;                  // forward launch, batches of more groups than one chip-load (round 4): groups per PASS -- a role's set s serves the groups of sets
                              // s, s + gpp / C, s + 2 gpp / C, ... one after the other (gpp is a multiple of every role's C); 0: all groups at once
    int* status;
    unsigned long long spin_ticks;
    unsigned poll_mask;
    const short* block_tab;   // [grid][4]: role, set, slice p, unused  (role < 0: the block leaves at once)
    int* census;              // residency check: every role-bearing workgroup signs in here first (zeroed with the flags) ...
    int n_active;             // ... and waits, briefly, until all n_active have: a launch that is not wholly resident (another process
    unsigned long long census_ticks;   // holds CUs) sets status 2 and leaves within census_ticks instead of spinning for seconds in its waits
    unsigned long long census_late_ticks;   // test hook (PAULE_HIP_DEBUG=census_late_ms=N): workgroup 0 signs in this late; 0 = off
    unsigned long long* stamps;
    int prio;                 // lstm_fused2.hip (round 5, PAULE_HIP_FUSED2_PRIO): roles of at most this many chains raise their waves' issue priority (s_setprio 3) -- the
                              // predictor's recurrence is the launch's dependent chain and shares its CU with a role that has slack; 0 = all waves equal
    const FusedRole* roles;   // [n_roles] in device memory (a table in the kernel arguments would have to be indexed dynamically,
                              // which makes the compiler copy it to scratch)
};
// hidden sizes (padded) the fused kernels are instantiated for
bool fused_supported(int Hp);
// forward launch: the predictor's roles (recurrences of all its layers, the projections between them, the mel head) and the
// embedder's may have different hidden sizes (model set B: 4 x 180 in front of 720)
bool fused_fwd_supported(int Hp_pred, int Hp_emb);
void launch_fused_fwd(hipStream_t stream, int Hp_pred, int Hp_emb, const FusedArgs& a);
void launch_fused_bwd(hipStream_t stream, int Hp_pred, int Hp_emb, const FusedArgs& a);
// the forward launch at TWO workgroups per CU (lstm_fused2.hip, round 4): the same role table, planned for 2 x n_cu workgroup slots
bool fused_fwd2_supported(int Hp_pred, int Hp_emb);
bool fused_passes_compiled();    // lstm_fused.hip was built with -DFUSED_PASSES=1 (the forward launch in passes, PAULE_HIP_FUSED_GPP)
bool fused_fwd2_xcd_compiled();   // lstm_fused2.hip was built with -DFUSED2_XCD=1 (the recurrence roles' own exchange through the XCD's L2)
void launch_fused_fwd2(hipStream_t stream, int Hp_pred, int Hp_emb, const FusedArgs& a);
// the two-per-CU recurrence role alone as a per-layer forward sweep (32-row groups, whole sequence; more groups than one pass of the
// one-per-CU sweep holds): sets = groups served at once
bool lstm_fwd2_sweep_supported(int Hp);
void launch_lstm_fwd2_sweep(hipStream_t stream, int Hp, int sets, const LstmSweepArgs& s);
// the same role tables with the LSTM roles on 16 batch rows (v_mfma_f32_16x16x32_bf16, one chain): batches of up to 16 rows
void launch_fused_fwd16(hipStream_t stream, int Hp_pred, int Hp_emb, const FusedArgs& a);
void launch_fused_bwd16(hipStream_t stream, int Hp_pred, int Hp_emb, const FusedArgs& a);
// the 16-row backward launch also takes a stacked predictor of one width in front of an embedder of another (model set B)
bool fused_bwd16_supported(int Hp_pred, int Hp_emb);
// zeroes n ints with write-through (sc1) stores: the arrival counters must not linger in any XCD's L2
void launch_zero_counters(hipStream_t stream, int* p, int n);

// ---- elementwise.hip ------------------------------------------------------------------------
// weight repack: src f32 [nblk*R, C] (torch layout) -> dst `dt` [nblk*Rp, Cp] (transpose = 0)
//                                              or -> dst `dt` [Cp, nblk*Rp] (transpose = 1), zero padded
void launch_pack_matrix(hipStream_t stream, int dt, const float* src, int nblk, int R, int C, void* dst, int Rp, int Cp,
                        bool transpose);
// bias: dst f32 [nblk*Rp] = b0 + b1 (b1 may be null), zero padded
void launch_pack_bias(hipStream_t stream, const float* b0, const float* b1, int nblk, int R, float* dst, int Rp);

// x master f64 [B,T,C] -> time-major activation [T][Bp][Cp] (zero padded)
void launch_pack_cp(hipStream_t stream, int dt, const double* x, int B, int T, int C, void* dst, int Bp, int Cp);
// user mel f32 [B,Tp,C] -> time-major activation [Tp][Bp][Cp]
void launch_pack_mel(hipStream_t stream, int dt, const float* mel, int B, int Tp, int C, void* dst, int Bp, int Cp);
// Y f32 [T][Bp][Cp] -> pooled mel: f32 batch-major [B][Tp][C] and activation time-major [Tp][Bp][Cp]
void launch_pool_mel(hipStream_t stream, int dt, const float* Y, int B, int T, int C, int Bp, int Cp, float* mel_bm,
                     void* mel_tm, int tp0 = 0, int n_tp = -1);   // n_tp >= 0: pooled frames tp0 .. tp0 + n_tp - 1 only
// gather rows h[lens[b]-1][b][:] (lens null -> Tl) of a time-major activation buffer into [Bp][Hp]
void launch_gather_last(hipStream_t stream, int dt, const void* h_tm, const int32_t* lens, int B, int Tl, int Bp, int Hp,
                        void* dst);

struct LossArgs {
    int B, T, Tp, C, M, S;           // batch, frames, mel frames, cp dim, mel dim, sem dim
    int Bp, Mp, Sp;                  // padded
    float w_mel, w_sem, w_vel, w_jerk, w_ll;
    int use_mel, use_sem;            // which terms enter the objective
    const double* x;                 // CP master [B][T][C]
    const float* mel;                // pred mel batch-major [B][Tp][M]
    const float* target_mel;         // [B][Tp][M]
    const float* sem;                // pred semvec f32 [Bp][Sp] (null if not evaluated)
    // somatosensory feedback (paule/paule.py:624-644, :739-757): the tube path's mel prediction [B][Tp][M] and semantic vector
    // [Bp][Sp] against the SAME targets, weights w_mel / w_sem (TUBE_MEL_WEIGHT = MEL_WEIGHT, :598-599); null = term off
    const float* mel2;
    const float* sem2;
    const float* target_sem;         // [B][S]
    double* dwork;                   // [B][3][T][C] velocity / jerk / local-linear correlations (kept for the gradient)
    double* part;                    // [B][loss_chunks(T)][8] partial sums of the reduction's workgroups (one per 256 CP frames of an utterance)
    double* scal;                    // per-utterance scalars [B][8]: 0 mel rmse, 1 sem rmse, 2 vel mse, 3 jerk mse, 4 ll mse, 5 classifier logit,
                                     // 6 tube-mel rmse, 7 tube-semvec rmse
    const float* cls_wb;             // speech classifier: [M] weights then bias, or null (term off)
    float w_cls;                     // its loss weight (0.1)
    float* loss_rows;                // [cap][B][8] internal log
    const int* iter_slot;            // device counter: row of loss_rows written by this iteration
};
// per-utterance reductions (deterministic, no atomics) -> scal
int loss_chunks(int T);   // workgroups per utterance of the loss reduction: a function of T only (an utterance's sums do not depend on its batch)
void launch_loss_reduce(hipStream_t stream, const LossArgs& a);
// writes loss_rows[*iter_slot][b][0..7]
void launch_loss_finalize(hipStream_t stream, const LossArgs& a);
// dsem activation [Bp][Sp] = w_sem * (sem - target) / (S * rmse_sem)   (zero rows for b >= B, zero pad)
// tube = true: the same for the tube embedder's vector (a.sem2, scalar slot 7)
void launch_dsem(hipStream_t stream, int dt, const LossArgs& a, void* dsem, bool tube = false);
// dY activation [T][Bp][Mp]: 0.5 * (use_mel * w_mel (mel - tgt)/(N rmse) + dmel_e[t/2][b][m]) ; dmel_e f32 [Tp][Bp][Mp] or null
// tube = true: the tube-mel model's output gradient: w_mel (mel2 - tgt)/(N rmse2) only (always part of the objective)
void launch_dy(hipStream_t stream, int dt, const LossArgs& a, const float* dmel_e, void* dY, bool tube = false, int t0 = 0, int n_t = -1);   // n_t >= 0: frames t0 .. only
// out (activation type) = a + b (f32), n elements: the two gradient streams that meet at the predicted tube
void launch_add2_act(hipStream_t stream, int dt, const float* a, const float* b, int64_t n, void* out);
// time-major padded f32 [T][Bp][Cp] -> batch-major [B][T][C]
void launch_tm_to_bm(hipStream_t stream, const float* src, int B, int T, int C, int Bp, int Cp, float* dst);

// grad = dX^T (+ dX2^T) + d(smoothness)/dx, then Adam + clamp + smiling + past_cp in place on x/m/v; bumps step_count and iter_slot
void launch_cp_update(hipStream_t stream, const AdamArgs& a);

// misc conversions
void launch_f64_to_f32(hipStream_t stream, const double* src, float* dst, int64_t n);
void launch_f32_to_f64(hipStream_t stream, const float* src, double* dst, int64_t n);
void launch_act_to_f32(hipStream_t stream, int dt, const void* src, float* dst, int64_t n);
// strip padding: src f32 [Bp][Sp] -> dst [B][S]
void launch_unpad_rows(hipStream_t stream, const float* src, int B, int S, int Sp, float* dst);

// ---- continued learning of the predictive model (train.hip) ------------------------------------------------------
// C[M][N] (f32) = sum over t < Tk, b < nb of A[(tA0 + t) * Bp + b][m] * B[(tB0 + t) * Bp + b][n]; nb multiple of 16, M, N of 8
// scratch (>= train_scratch_bytes of the largest product) holds split-K partials: a block's K loop is serial, so small
// outputs are split over K to fill the chip and summed in a fixed order afterwards
size_t train_scratch_bytes(int M, int N);
void launch_gemm_tn(hipStream_t stream, int dt, const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N,
                    int Bp, int nb, int Tk, int tA0, int tB0, float* scratch, size_t scratch_bytes, int n_cu, bool tn_bf16 = true);
// out[c] = sum over the same rows of A[.][c]; part: >= 64 * ncols doubles
void launch_colsum(hipStream_t stream, int dt, const void* A, int lda, int ncols, int Bp, int nb, int Tk, int t0, float* out, double* part);
// scal[0] = sqrt(mean((pred - target)^2)) over n elements (f64), scal[1] = the sum of squares; loss_out (device f32) optional
void launch_train_rmse(hipStream_t stream, const float* pred, const float* target, int64_t n, double* scal, float* loss_out);
void launch_train_dy(hipStream_t stream, int dt, const float* pred, const float* target, const double* scal, int n_rows, int T, int Tp,
                     int M, int Bp, int Mp, void* dY, bool pooled = true);
struct AdamHyper {
    double lr, b1, b2, eps, bc1, bc2;   // bc = 1 - beta^k of the step being taken
};
void launch_adam_matrix(hipStream_t stream, int dt, const float* grad, int nblk, int R, int C, int Rp, int Cp, double* x, double* am,
                        double* av, void* W, void* WT, const AdamHyper& hp);
void launch_adam_bias(hipStream_t stream, const float* grad, int nblk, int R, int Rp, double* x0, double* am0, double* av0, double* x1,
                      double* am1, double* av1, float* packed, const AdamHyper& hp);

// ---- inverse model forward (inverse.hip; InverseModelMelTimeSmoothResidual, paule/models.py:177-247) ---------------
void launch_mel_block(hipStream_t st, const float* x, int B, int Tp, int M, const float* w, const float* b, float* y);
// backward-data of a mel block; element (b, t, c) at b * sb + t * st + c on either side (batch-major or time-major padded)
void launch_mel_block_bwd(hipStream_t st, const float* dy, int64_t sb_in, int64_t st_in, int B, int Tp, int M, const float* w, float* dx,
                          int64_t sb_out, int64_t st_out);
// LeakyReLU of the embedder head and its backward (by the sign of the pre-activation)
void launch_leaky(hipStream_t st, int dt, const float* pre, int64_t n, float slope, void* out);
void launch_leaky_bwd(hipStream_t st, int dt, const float* d, const float* pre, int64_t n, float slope, void* out);
void launch_vel_acc_pack(hipStream_t st, int dt, const float* x, int B, int Tp, int M, void* dst, int Bp, int in_p);
void launch_double_seq(hipStream_t st, const float* Y, int B, int Tp, int C, int Bp, int Cp, float* z);
void launch_time_conv5(hipStream_t st, const float* x, int B, int T, int C, const float* w, const float* b, const float* resid, float* y);
void launch_resid_weight(hipStream_t st, const float* zs, const float* zl, int B, int T, int C, const float* w, const float* b, int clip,
                         float* y);
void launch_clip_copy(hipStream_t st, const float* x, int64_t n, int clip, float* y);

}  // namespace pl
