// Planner handle, per-iteration launch sequence and the C-ABI of include/paule_hip.h.
//
// One inner iteration (paule/paule.py:910-1211 minus the log-step block) is the fixed kernel sequence
// built by enqueue_iteration(); it is captured once into a hipGraph and replayed n_iters times by
// pl_step.  Everything that varies between iterations (Adam step count, loss-log row) lives in device
// counters, so the captured graph is static.
#include <hip/hip_runtime.h>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <unordered_map>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <atomic>
#include <vector>

#include "../../include/paule_hip.h"
#include "kernels.h"
#include "pl_types.h"

using namespace pl;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define PL_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(PL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));               \
    } while (0)

// A persistent sweep needs all workgroups of a batch group resident at once; two sweeps of DIFFERENT handles running
// concurrently on one device (two streams) could each hold CUs the other one's partners need, and both would spin into
// their bounded timeouts.  Every entry point that launches sweeps therefore chains its work behind the previous such
// call on the same device (one event per device, submission serialised by a mutex): handles on different streams of one
// process execute their sweeps one after the other.  Same stream: a no-op.  (Processes sharing one GPU are not covered.)
struct SweepChain {
    static constexpr int kMaxDev = 64;
    static std::mutex& mu(int dev) { static std::mutex m[kMaxDev]; return m[dev]; }
    static hipEvent_t& evt(int dev) { static hipEvent_t e[kMaxDev] = {}; return e[dev]; }
    int dev;
    hipStream_t stream;
    bool on;
    SweepChain(int device, hipStream_t st) : dev(device), stream(st), on(device >= 0 && device < kMaxDev) {
        if (!on) return;
        mu(dev).lock();
        if (evt(dev)) (void)hipStreamWaitEvent(stream, evt(dev), 0);
    }
    ~SweepChain() {
        if (!on) return;
        if (!evt(dev)) (void)hipEventCreateWithFlags(&evt(dev), hipEventDisableTiming);
        if (evt(dev)) (void)hipEventRecord(evt(dev), stream);
        mu(dev).unlock();
    }
};

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

// one trainable parameter tensor in torch layout: f64 master + Adam moments (moments allocated with the first training step)
struct ParamState {
    double *x = nullptr, *m = nullptr, *v = nullptr;
    size_t n = 0;
};

struct LstmLayer {
    int in = 0, in_p = 0;
    void *Wih = nullptr, *WihT = nullptr, *Whh = nullptr, *WhhT = nullptr;
    float* bias = nullptr;
    void *G = nullptr, *h = nullptr, *c = nullptr;   // time-major slabs [Tl][Bp][4Hp] / [Tl][Bp][Hp]
    bool set = false;
    ParamState p_wih, p_whh, p_bih, p_bhh;           // masters (continued learning, pl_get_lstm_weights)
    float *gWih = nullptr, *gWhh = nullptr, *gb = nullptr;   // weight gradients in the padded compute layout
    // layer wavefront (small batches): f32 cell state crossing a time-chunk border (forward c / backward dL/dc) and this
    // layer's own reduce-scatter exchange (the layers' backward sweeps run side by side)
    float *carry_f = nullptr, *carry_b = nullptr;
    void* xchg = nullptr;
};

struct Model {
    int L = 0, H = 0, Hp = 0, in = 0, in_p = 0, out = 0, out_p = 0, Tl = 0;
    std::vector<LstmLayer> layers;
    void *Wlin = nullptr, *WlinT = nullptr;   // [out_p][Hp], [Hp][out_p]
    float* blin = nullptr;
    void* dh_ext = nullptr;                   // [Tl][Bp][Hp] dL/dh of the layer being back-propagated
    bool lin_set = false;
    ParamState p_wlin, p_blin;
    float *gWlin = nullptr, *gblin = nullptr;
    float* train_scratch = nullptr;           // split-K partials of the weight-gradient products
    size_t train_scratch_bytes = 0;
    double* colsum_part = nullptr;
    bool train_ready = false;                 // moments + gradient buffers allocated
    long long train_steps = 0;                // Adam step count of the parameter optimizer
    bool ready() const {
        if (!lin_set) return false;
        for (auto& l : layers)
            if (!l.set) return false;
        return true;
    }
};

}  // namespace

// ONE diagnostic variable (round 5; VERDICT r4 hygiene): PAULE_HIP_DEBUG = comma-separated words out of
//   stop_after_fwd        pl_step enqueues the forward pass only (tests that compare forward stashes, tools/fused_check.py, tools/fused2_stamps.py)
//   fused                 print grid, expected workgroups and chain counts of every fused launch, and where the 16-row roles' workgroups sat
//   graph                 trace capture / instantiate / launch / destroy of the iteration graph on stderr
//   segv_trace            print the native frames of a segmentation fault before dying (section 10 of DESIGN.md)
//   census_ms=N           how long a fused launch's workgroups wait for each other to sign in (default 50)
//   census_late_ms=N      test hook: workgroup 0 signs in N ms late          } the two ways a census fails, made to happen
//   census_expect_extra=N test hook: the census waits for N more workgroups  } (tests/test_hip_parity.py)
// It replaces PAULE_HIP_STOP_AFTER_FWD, _DEBUG_FUSED, _DEBUG_GRAPH, _SEGV_TRACE and the three PAULE_HIP_CENSUS_* variables.
static const char* debug_find(const char* w) {   // -> what follows the word ("" for a bare word, "=..." for a value), or nullptr
    const char* v = std::getenv("PAULE_HIP_DEBUG");
    if (!v) return nullptr;
    const size_t n = std::strlen(w);
    for (const char* q = v; *q;) {
        const char* e = std::strchr(q, ',');
        const size_t len = e ? (size_t)(e - q) : std::strlen(q);
        if (len >= n && std::strncmp(q, w, n) == 0 && (len == n || q[n] == '=')) return q + n;
        if (!e) break;
        q = e + 1;
    }
    return nullptr;
}
static bool debug_word(const char* w) { return debug_find(w) != nullptr; }
static bool debug_value(const char* w, long long* out) {
    const char* r = debug_find(w);
    if (!r || *r != '=') return false;
    *out = std::atoll(r + 1);
    return true;
}

struct pl_handle {
    pl_config cfg{};
    int B = 0, T = 0, Tp = 0, C = 0, M = 0, S = 0, Bp = 0, Cp = 0, Mp = 0, Sp = 0, dt = 0;
    hipStream_t stream = nullptr;
    Model pred, emb;
    Model inv;                  // optional inverse model (initialisation from the target acoustics)
    int inv_mel_blocks = 0, inv_res_blocks = 0;
    float* inv_conv = nullptr;  // conv parameters: mel blocks [3][G][15] + [3][G] each, res convs [C][5] + [C] each, resid_weighting [C][10] + [C]
    std::vector<char> inv_conv_set;
    float* inv_x[2] = {nullptr, nullptr};      // [B][Tp][M] ping-pong of the mel blocks
    void* inv_in = nullptr;                     // [Tp][Bp][pad32(3M)] LSTM input
    float* inv_Y = nullptr;                     // [Tp][Bp][Cp] post_linear output
    float* inv_z[3] = {nullptr, nullptr, nullptr};   // [B][2Tp][C]: lstm_output, smoothed, scratch
    // somatosensory feedback (pl_config.cp_tube_layers > 0): CP -> tube, tube -> mel, tube -> semantic vector
    Model tube, tmel, temb;
    int U = 0, Up = 0;                 // tube_dim, padded
    void* tube_tm = nullptr;           // pred_tube, time-major activation [T][Bp][Up]: input of tmel and temb
    float* Y2 = nullptr;               // tube-mel post_linear output [T][Bp][Mp]
    float* mel2_bm = nullptr;          // pooled pred_tube_mel [Bp][Tp][M]
    void* mel2_tm = nullptr;           // (written by the pooling kernel, unused)
    void* h_last2 = nullptr;
    float* sem2 = nullptr;             // pred_tube_semvec [Bp][Sp]
    void *dsem2 = nullptr, *dv2 = nullptr, *dY2 = nullptr, *dYt = nullptr;
    float *dtube_a = nullptr, *dtube_b = nullptr;   // dL/dpred_tube through the tube embedder / the tube-mel model [T][Bp][Up]
    float* dX2 = nullptr;              // dL/dCP through the CP -> tube model [T][Bp][Cp]
    bool tube_on() const { return tube.L > 0; }
    // embedder variants (pl_config.emb_post_size / emb_mel_blocks): head post_linear -> LeakyReLU -> output mapping, mel blocks in front
    int emb_post = 0, emb_post_p = 0, emb_blocks = 0;
    void *Wup = nullptr, *WupT = nullptr;       // output mapping [Sp][post_p] and its transpose [post_p][Sp]
    float* bup = nullptr;
    bool up_set = false;
    float* emb_conv = nullptr;                  // mel blocks: [3][G][15] + [3][G] each
    std::vector<char> emb_conv_set;
    float* emb_x[2] = {nullptr, nullptr};       // [B][Tp][M] ping-pong of the mel blocks (forward and backward)
    void* emb_in = nullptr;                     // [Tp][Bp][Mp] LSTM input behind the mel blocks
    float* emb_din = nullptr;                   // [Tp][Bp][Mp] input gradient of the LSTM stack (time-major)
    float* head_pre = nullptr;                  // [Bp][post_p] post_linear output (pre-activation)
    void* head_act = nullptr;                   // [Bp][post_p] LeakyReLU(head_pre) / its gradient, activation type
    float* head_d = nullptr;                    // [Bp][post_p] gradient w.r.t. the activation
    float* c_run[2] = {nullptr, nullptr};
    float* dc_run[2] = {nullptr, nullptr};
    void* X0 = nullptr;
    float* Y = nullptr;
    float* mel_bm = nullptr;
    void* mel_tm = nullptr;
    void* h_last = nullptr;
    float* sem = nullptr;
    void* dsem = nullptr;
    void* dv = nullptr;
    float* dmel_e = nullptr;
    void* dY = nullptr;
    float* dX = nullptr;
    double *x = nullptr, *m = nullptr, *v = nullptr, *grad = nullptr, *dwork = nullptr;
    float *target_mel = nullptr, *target_sem = nullptr;
    float* cls_wb = nullptr;    // speech classifier weights + bias
    bool cls_on = false;
    float w_cls = 0.1f;
    float* out_tmp = nullptr;   // [B][max(S, ...)] staging for unpadded outputs
    double* scal = nullptr;
    double* loss_part = nullptr;   // [B][loss_chunks(T)][8]: partial sums of the loss reduction
    float* loss_rows = nullptr;
    int loss_cap = 0;
    int* counters = nullptr;    // [0] Adam step count, [1] loss row of the running iteration
    int* sweep_cnt = nullptr;   // arrival flags of the persistent LSTM sweeps: n_sweep_slots slices of [groups][T][flag_stride] + XCD table
    size_t sweep_cnt_bytes = 0; // bytes of ONE slice
    int n_sweep_slots = 1;      // one slice per sweep of an iteration, so that ONE launch zeroes them all
    int sweep_slot = -1;        // >= 0 while an iteration is being enqueued: next slice to use (already zeroed)
    int* sweep_status = nullptr;
    int n_cu = 0;
    int flag_stride = 16;
    bool fuse_input = true;     // fuse narrow input projections into the forward sweeps
    bool use_sweep = true;
    int zero_mode = 0;          // 0: own sc1 zeroing kernel, 1: hipMemsetAsync (experiments)
    int xcd_fast = 2;           // same-XCD groups hand off through the shared L2 (verified at run time): bit 0 forward, bit 1 backward
                                // sweeps.  Default backward only: its hand-off is whole 128-byte lines (-4 % iteration time); the forward
                                // hand-off is half lines per workgroup and reads back slower from L2 than from the memory side (+1 %)
    int bwd_mode = 1;           // backward sweep: 1 reduce-scatter of partial dh tiles (default: 8.5 % faster iteration with the
                                // same-XCD fast path), 0 all-gather of dA (f32-exact accumulation; A/B variant)
    // diagnostic switches, read ONCE by pl_create (no getenv on launch paths)
    bool tn_bf16 = true;        // PAULE_HIP_TN_BF16: training, bf16: weight-gradient products on the bf16 MFMA (0: the exact f32 MFMA form)
    bool stop_after_fwd = false;   // PAULE_HIP_DEBUG=stop_after_fwd: pl_step enqueues the forward pass only (tools/fused_check.py, stash comparisons in tests)
    bool debug_fused = false;   // PAULE_HIP_DEBUG=fused
    bool gemm_big = true;       // PAULE_HIP_GEMM_BIG: the large bf16 products on gemm_big.hip's 256 x 256 tiles (0: gemm.hip's tiles; same bits)
    bool census_hooks = false;  // PAULE_HIP_DEBUG=census_expect_extra / census_late_ms were set when the handle was created: the test
                                // hooks of the residency census are then re-read at every fused launch (never otherwise)
    int bwd_stream = 1;         // PAULE_HIP_BWD_STREAM: form of the 32-row reduce-scatter backward sweep's hand-off (lstm_persist_rs.hip): 2 the tiles
                                // carry their own step token (round 4: no flags, no drains), 1 per-tile flags and streamed ingest (round 3), 0 one
                                // whole-workgroup hand-off per step
    int bwd_waves = 8;          // PAULE_HIP_BWD_WAVES: waves per workgroup of the reduce-scatter backward sweep (4: one per SIMD, round 2's form)
    int n_layer_slots_zeroed = 0;   // per-layer flag slices zeroed at the top of the running iteration (take_sweep_slice hands out no other)
    void* sweep_xchg = nullptr; // exchange buffer of the reduce-scatter backward sweep
    int bwd_dma = 0;              // PAULE_HIP_BWD_DMA (A/B): the streamed backward sweep fetches its stash rows a step ahead by LDS-DMA (bit-identical)
    bool token_early = true;          // PAULE_HIP_TOKEN_EARLY (A/B): the token form loads first / second-round tiles during its own tile phase
    void* sweep_xchg_tok = nullptr;   // ... of its token form, which nobody else may write: all zero ("retired") between launches
    size_t sweep_xchg_tok_bytes = 0;
    bool f32_sweep = true;      // PAULE_HIP_F32_SWEEP: persistent sweeps on the f32 path
    int f32_chains = -1;        // PAULE_HIP_F32_CHAINS: f32 batches of more groups than fit the chip: -1 auto, 0 off (groups take turns / launch-per-step), N forced
    bool stash_lds = true;      // PAULE_HIP_STASH_LDS: forward stash stores staged through LDS (whole 64-byte row pieces)
    bool xcd_fast16 = true;     // PAULE_HIP_XCD_FAST16: forward same-XCD hand-off for the 16-row kernels
    bool own_store = true;      // PAULE_HIP_OWN_STORE: backward 32-row kernel: every wave hands its own partial tiles over behind their MFMAs
    bool wide_ingest = true;    // PAULE_HIP_WIDE_INGEST: f32 backward sweep sums the partial tiles with 16-byte loads, wave by wave
    bool wide_ingest16 = true;  // PAULE_HIP_WIDE_INGEST16: 16-row bf16 backward sweep sums the partial tiles with 16-byte loads, wave by wave
    bool pipe_spread = true;    // PAULE_HIP_PIPE_SPREAD: pipelines launch small bf16 16-row sweeps spread over the XCDs (pipe_spread16)
    bool wf_pipeline = true;    // PAULE_HIP_WF_PIPELINE: predictor -> mel head -> embedder as one pipeline over time chunks (small batches)
    bool f32_valu = true;       // PAULE_HIP_F32_VALU: f32 sweeps of at most 4 rows in use run their recurrent products as FMA chains
    int rows_in_use = 0;        // batch rows that carry data (B); 0 inside pl_train_model_step (few-row kernels off)
    int wavefront = 4;          // PAULE_HIP_WAVEFRONT = time chunks (0 off): small sweeps run the layers of a model side by side, layer l + 1
                                // on chunk c while layer l is on chunk c + 1 (model_forward_wavefront)
    int wf_dirs = 3;            // PAULE_HIP_WF_DIRS: bit 0 forward, bit 1 backward (experiments)
    // graph mode: the wavefront is captured on ONE stream (a chain of nodes) and its edges are rewritten before the graph is
    // instantiated (build_graph): multi-stream capture crashed inside hipStreamEndCapture about once in a hundred captures
    // (tools/microbench/capture_stress.py), single-stream capture never did
    struct WfSeg { hipGraphNode_t before, last; int dep1, dep2; };   // chain neighbours of a (layer, chunk) segment; true dependencies
    struct WfRegion { std::vector<WfSeg> segs; std::vector<int> finals; };
    std::vector<WfRegion> wf_regions;
    bool capturing = false;
    size_t wf_stream_next = 0;  // cursor into the device's pool of layer streams: every wavefront region of an iteration takes its own
                                // (HIP's capture does not survive a stream that is forked, joined and forked again with others)
    bool wf_on = false;
    size_t wf_next = 0;         // cursor into the device's pool of events ((layer, chunk) dependencies: none is recorded twice inside one iteration)
    bool sweep16 = true;        // PAULE_HIP_SWEEP16: 16-row groups for bf16 batches of up to 128 rows (lstm_persist16.hip)
    bool small_grid = true;     // PAULE_HIP_SMALL_GRID: batches of fewer than 8 groups still launch 8 group slots, which keeps each
                                // group on one XCD (B = 8: 5.40 -> 4.98 ms per iteration, profiles/r01_ab_small_batch_grid.txt)
    unsigned long long* sweep_stamps = nullptr;   // -DPL_STAMPS builds: [2 (fwd/bwd)][256 blocks][8]
    // fused acoustic sweeps (lstm_fused.hip): one persistent launch per direction, workgroups take roles from these tables
    bool fused_fwd_ok = false;  // PAULE_HIP_FUSED bit 0 and the shapes / CU budget fit (plan_fused)
    int fused_Cp = 0, fused_Ce = 0;   // chains per workgroup of the predictor's / the embedder's roles: forward launch
    int fused_Cp_bwd = 0, fused_Ce_bwd = 0;   // ... backward launch (the same unless the forward launch runs two workgroups per CU)
    int f32_stream = 0;   // PAULE_HIP_F32_STREAM: 1 = per-tile hand-off in the whole-sequence f32 backward sweeps (lstm_bwd_stream_f32_kernel; opt-in: bit-identical, measured neutral -- profiles/r05_f32_stream.txt), 0 = one flag per workgroup and step
    int fused16_pf = 2;    // PAULE_HIP_FUSED16_PF: stash prefetcher workgroups behind every recurrence set of the 16-row fused backward launch (0: none; cfg5 13.72 -> 13.54 ms with 2, 4 or 8)
    int fused2_prio = -1;  // PAULE_HIP_FUSED2_PRIO (FusedArgs::prio): -1 auto = 1 where a one-layer predictor runs on ONE chain beside embedder roles on two or more (set A at
                           // 193 ... 256 rows: its 300 chain-steps are the launch; cfg3 4.088 -> 4.052 ms), else 0 (set B: 3.99 -> 4.02 with it); profiles/r05_ab_fused2_prio.txt
    int bwd_pf = -1, bwd_pf_dist = 4;   // PAULE_HIP_BWD_PF / _PF_DIST: stash prefetcher workgroups of the streamed backward sweeps (LstmSweepArgs::n_pf): -1 auto, 0 off
    int bwd_chains = 0;           // PAULE_HIP_BWD_CHAINS: > 0: the 32-row streamed backward sweeps in chained form, that many groups per workgroup (lstm_bwd_rs_chain_kernel)
    int bwd_xt = 2;               // PAULE_HIP_BWD_XT: 2 (default) = the predictor's input gradient rides along in its streamed backward sweep (lstm_persist_rs.hip, XT) and the
                                  // sweep does not write the layer's dA over the gate stash (nothing in a planning iteration reads it once dL/dCP comes from the ride-along
                                  // tile: 454 MB of stores per launch at cfg3, 4.10 -> 4.04 ms); 1 = ride-along, dA kept (pl_debug_read of pred.G0 behind a step); 0 = dL/dCP by the batched product
    bool pred_dA_skipped = false; // the last planning iteration left the predictor's first-layer gate stash as the forward pass wrote it (pl_debug_read refuses to pass it off as dA)
    float* dx_part = nullptr;         // its scratch: the workgroups' partial tiles, f32 [T][groups][P][32 x 32]
    bool fused2_xcd = false;          // PAULE_HIP_FUSED2_XCD=1: the two-per-CU forward roles' own exchange through the XCD's L2 from a private tile-major copy (round 5:
                                      // bit-identical; the predictor's chain-step 5.3 -> 5.1 us, the embedder's roles -- the launch's longest -- unchanged, the iteration
                                      // 4.35 -> 4.39 ms: opt-in, profiles/r05_ab_fused2_forward.txt)
    bool fused_xcd = true;            // PAULE_HIP_FUSED_XCD: the 32-row fused backward roles' own exchange through the shared L2 when a set sits on one XCD
    int fused_gpp = 0;                // forward launch in passes: groups per pass (0: every group has its own set, one pass)
    short* fused_tab_fwd = nullptr;   // [n_cu][4] block -> (role, set, slice)
    int fused_grid_fwd = 0;
    bool fused_bwd_ok = false;  // PAULE_HIP_FUSED bit 1
    int sweep2 = -1;             // PAULE_HIP_SWEEP2: per-layer forward sweeps of batches beyond one pass on the two-per-CU role (1 always, 0 never, -1 auto)
    int fused_occ2 = -1;         // PAULE_HIP_FUSED_OCC2: the forward launch at two workgroups per CU (lstm_fused2.hip): 1 wherever the shape has it, 0 never, -1 (default) where it wins (plan_fused)
    bool fused_fwd2 = false;     // ... and the plan took it: the role table holds up to 2 n_cu workgroups
    bool fused_rows16 = false;  // batches of up to 16 rows: the LSTM roles of both launches run on 16-row tiles (lstm_fused16.h)
    void* fused_hx[kFusedMaxRoles] = {};        // rows16: the forward LSTM roles' own copies of their h hand-off [2][16][Hp]
    void* fused_dh_pred[4] = {};                // stacked predictor in the backward launch: dL/dh rows of layer l < L - 1 (written by the product role of layer l + 1)
    short* fused_tab_bwd = nullptr;
    int fused_grid_bwd = 0;
    void* fused_xchg[kFusedMaxRoles] = {};      // recurrence exchange of every backward LSTM role (they run side by side)
    void* fused_xchg_ext[kFusedMaxRoles] = {};  // ring of dL/dh partial tiles written by the FR_DX_BWD role with this index
    void* fused_xchg_mel = nullptr;             // ring of the embedder's input-gradient partial tiles
    FusedRole* fused_roles_fwd = nullptr;       // role tables in device memory (static for the life of the handle: every pointer is the
    FusedRole* fused_roles_bwd = nullptr;       // handle's own buffer, the flag slices are the last 2 n_roles of the iteration's slices)
    int fused_n_roles = 0;
    int fused_active_fwd = 0, fused_active_bwd = 0;   // role-bearing workgroups of the two launches (residency census)
    unsigned long long census_ticks = 5000000ull;     // 50 ms (PAULE_HIP_DEBUG=census_ms=N)
    unsigned long long spin_ticks = 200000000ull;   // 2 s
    unsigned poll_mask = 63u;
    double* past = nullptr;
    int past_len = 0, past_per_utt = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t cap_stream = nullptr;
    bool have_targets = false, have_sem_target = false, have_cp = false;
    size_t bytes = 0;
    std::vector<void*> allocs;
    size_t act = 4;   // bytes per activation element

    bool need_emb_in_step() const { return cfg.objective != PL_OBJ_ACOUSTIC; }
    bool use_mel() const { return cfg.objective != PL_OBJ_SEMVEC; }
};

namespace {

int raw_alloc(pl_handle* h, void** p, size_t nb) {
    void* q = nullptr;
    PL_HIP(hipMalloc(&q, nb ? nb : 16));
    PL_HIP(hipMemsetAsync(q, 0, nb ? nb : 16, h->stream));
    h->allocs.push_back(q);
    h->bytes += nb;
    *p = q;
    return PL_OK;
}

template <typename T>
int dev_alloc(pl_handle* h, T** p, size_t n_elems) {
    void* q = nullptr;
    int rc = raw_alloc(h, &q, n_elems * sizeof(T));
    *p = static_cast<T*>(q);
    return rc;
}

int alloc_act(pl_handle* h, void** p, size_t n_elems) { return raw_alloc(h, p, n_elems * h->act); }

int alloc_model(pl_handle* h, Model& md, int L, int H, int in, int out, int Tl) {
    md.L = L;
    md.H = H;
    md.Hp = pad32(H);
    md.in = in;
    md.in_p = pad32(in);
    md.out = out;
    md.out_p = pad32(out);
    md.Tl = Tl;
    md.layers.resize(L);
    const size_t Bp = h->Bp, Hp = md.Hp;
    for (int l = 0; l < L; ++l) {
        LstmLayer& ly = md.layers[l];
        ly.in = l == 0 ? in : H;
        ly.in_p = pad32(ly.in);
        int rc;
        if ((rc = alloc_act(h, &ly.Wih, 4 * Hp * ly.in_p))) return rc;
        if ((rc = alloc_act(h, &ly.WihT, (size_t)ly.in_p * 4 * Hp))) return rc;
        if ((rc = alloc_act(h, &ly.Whh, 4 * Hp * Hp))) return rc;
        if ((rc = alloc_act(h, &ly.WhhT, Hp * 4 * Hp))) return rc;
        if ((rc = dev_alloc(h, &ly.bias, 4 * Hp))) return rc;
        if ((rc = alloc_act(h, &ly.G, (size_t)Tl * Bp * 4 * Hp))) return rc;
        if ((rc = alloc_act(h, &ly.h, (size_t)Tl * Bp * Hp))) return rc;
        if ((rc = alloc_act(h, &ly.c, (size_t)Tl * Bp * Hp))) return rc;
        ly.p_wih.n = (size_t)4 * H * ly.in;
        ly.p_whh.n = (size_t)4 * H * H;
        ly.p_bih.n = ly.p_bhh.n = (size_t)4 * H;
        for (ParamState* ps : {&ly.p_wih, &ly.p_whh, &ly.p_bih, &ly.p_bhh})
            if ((rc = dev_alloc(h, &ps->x, ps->n))) return rc;
    }
    md.p_wlin.n = (size_t)out * H;
    md.p_blin.n = (size_t)out;
    for (ParamState* ps : {&md.p_wlin, &md.p_blin}) {
        int rc2;
        if ((rc2 = dev_alloc(h, &ps->x, ps->n))) return rc2;
    }
    int rc;
    if ((rc = alloc_act(h, &md.Wlin, (size_t)md.out_p * Hp))) return rc;
    if ((rc = alloc_act(h, &md.WlinT, Hp * md.out_p))) return rc;
    if ((rc = dev_alloc(h, &md.blin, md.out_p))) return rc;
    if ((rc = alloc_act(h, &md.dh_ext, (size_t)Tl * Bp * Hp))) return rc;
    return PL_OK;
}

inline char* off(void* p, size_t elems, size_t esz) { return static_cast<char*>(p) + elems * esz; }

// flags of the next sweep: inside an iteration the slices were zeroed together up front (zero_all_sweep_slots) and are handed
// out in turn; elsewhere (forward-only calls, training, timing) slice 0 is zeroed right before the sweep
int* take_sweep_slice(pl_handle* h, hipStream_t st) {
    const size_t ints = h->sweep_cnt_bytes / sizeof(int);
    // only slices that zero_all_sweep_slots zeroed at the top of this iteration AND that no fused role owns are handed out (a
    // stale non-zero flag makes every in-kernel wait fall through); beyond them slice 0 is zeroed on the spot, as outside an iteration
    if (h->sweep_slot >= 0 && h->sweep_slot < h->n_layer_slots_zeroed) return h->sweep_cnt + (size_t)(h->sweep_slot++) * ints;
    if (h->sweep_slot >= 0) h->sweep_slot = h->n_sweep_slots;   // (stays past the bound for the rest of the iteration)
    if (h->zero_mode == 1)
        (void)hipMemsetAsync(h->sweep_cnt, 0, h->sweep_cnt_bytes, st);
    else
        launch_zero_counters(st, h->sweep_cnt, (int)ints);
    return h->sweep_cnt;
}
void zero_all_sweep_slots(pl_handle* h, hipStream_t st) {
    // the per-layer slices come first, then the flag slices of the fused forward launch's roles, then the fused backward launch's:
    // only what this handle's plan hands out is zeroed (a handle without fused launches does not pay for their slices: ADVICE r2)
    int n_used = h->n_sweep_slots, n_layer = h->n_sweep_slots;
    if (h->cfg.emb_layers > 0) {
        const int n_roles = (2 * h->cfg.pred_layers - 1) + 1 + (2 * h->cfg.emb_layers - 1), n_fused = 2 * n_roles + (h->cfg.emb_layers - 1) + (h->cfg.pred_layers - 1);
        n_used = h->n_sweep_slots - n_fused;
        n_layer = n_used;
        if (h->fused_bwd_ok) n_used = h->n_sweep_slots;
        else if (h->fused_fwd_ok) n_used += n_roles;
    }
    const size_t ints = h->sweep_cnt_bytes / sizeof(int) * n_used;
    if (h->zero_mode == 1)
        (void)hipMemsetAsync(h->sweep_cnt, 0, ints * sizeof(int), st);
    else
        launch_zero_counters(st, h->sweep_cnt, (int)ints);
    h->n_layer_slots_zeroed = n_layer;   // the per-layer slices come first; what follows belongs to the fused launches' roles
    h->sweep_slot = 0;
}

// persistent-sweep dispatch by arithmetic type: grid 0 = launch-per-step kernels
// 16-row groups (lstm_persist16.hip) for small bf16 batches; the backward needs the reduce-scatter exchange
bool use_sweep16(pl_handle* h, int Hp, bool bwd) {
    if (h->dt != BF16 || !h->sweep16 || !lstm_sweep_supported(h->dt, Hp) || !lstm_sweep16_wanted(Hp, h->Bp, h->n_cu)) return false;
    return !bwd || (h->bwd_mode == 1 && h->sweep_xchg);
}
// f32 batches whose 16-row groups do not all fit the chip: chains (lstm_chain_f32.hip) instead of groups taking turns.
// PAULE_HIP_F32_CHAINS: 0 off, N > 0 force N chains per workgroup (tests), unset: as many as make every group resident
int f32_chains_for(pl_handle* h, int Hp, int* grid) {
    *grid = 0;
    if (h->f32_chains == 0 || h->dt != F32) return 0;
    return lstm_chain_f32_plan(Hp, h->Bp, h->n_cu, h->f32_chains > 0 ? h->f32_chains : 0, grid);
}
int sweep_grid_for(pl_handle* h, int Hp, bool bwd = false) {
    if (!h->use_sweep) return 0;
    if (h->dt == F32) {
        if (!h->f32_sweep || !lstm_sweep_f32_supported(Hp)) return 0;
        int cgrid = 0;
        if (f32_chains_for(h, Hp, &cgrid)) return cgrid;
        return lstm_sweep_f32_grid(Hp, h->Bp, h->n_cu);
    }
    if (use_sweep16(h, Hp, bwd)) return lstm_sweep16_grid(Hp, h->Bp, h->n_cu, h->small_grid);
    return lstm_sweep_supported(h->dt, Hp) ? lstm_sweep_grid(Hp, h->Bp, h->n_cu, h->small_grid) : 0;
}
int sweep_group_rows_for(pl_handle* h, int Hp, bool bwd = false) {
    if (h->dt == F32 || use_sweep16(h, Hp, bwd)) return 16;
    return lstm_sweep_group_rows(Hp, h->Bp, h->n_cu);
}
// Prefetcher workgroups beside a streamed 32-row backward sweep of `grid` workgroups (round 5): as many per resident group as the idle CUs
// hold, at most 8 (cfg3: 8 groups x 23 workgroups leave 72 CUs; 4 per group bring 4.99 -> 4.70 ms per iteration, 8 -> 4.66, 9 no more:
// profiles/r05_ab_prefetchers.txt).  PAULE_HIP_BWD_PF: -1 auto, 0 off, n per group at most.
int bwd_prefetchers_for(const pl_handle* h, int Hp, int grid, int slice = 32) {   // slice: hidden units per workgroup (32 bf16, 16 f32)
    const int P = Hp / slice;
    if (h->bwd_pf == 0 || P < 1 || grid < P) return 0;
    const int n_res = grid / P;
    int per = h->bwd_pf > 0 ? h->bwd_pf : 8;
    while (per > 0 && grid + per * n_res > h->n_cu) --per;
    return per * n_res;
}

void launch_sweep(pl_handle* h, hipStream_t st, bool bwd, int Hp, int grid, const LstmSweepArgs& s) {
    if (h->dt == F32) {
        int cgrid = 0;
        const int C = f32_chains_for(h, Hp, &cgrid);
        if (C > 0 && s.t0 == 0 && (s.t1 == 0 || s.t1 == s.T)) {
            LstmSweepArgs sc = s;
            sc.chains = C;
            if (bwd) { sc.n_pf = bwd_prefetchers_for(h, Hp, cgrid, 16); sc.pf_dist = h->bwd_pf_dist; }
            launch_lstm_chain_f32(st, bwd, Hp, cgrid, sc);
        } else {
            // PAULE_HIP_F32_STREAM=1 (round 5, opt-in): whole-sequence backward sweeps on the MFMAs with the streamed per-tile hand-off;
            // its tile flags sit behind the XCD-id table of the same flag slice, as in bf16
            LstmSweepArgs sf = s;
            if (bwd && h->f32_stream && s.t0 == 0 && (s.t1 == 0 || s.t1 == s.T)) sf.tflags = s.xcc_tab + (size_t)((h->Bp + 7) / 8) * 64;
            // (no prefetchers beside the one-chain f32 sweeps: their step is MFMA issue, 3.1 of 6.0 us, and the flag wait 0.8 -- cfg2 3.88 ms
            // per iteration with and without, profiles/r05_ab_prefetchers.txt; the chains kernel above gains 4 %)
            launch_lstm_sweep_f32(st, bwd, Hp, grid, sf);
        }
    }
    else if (use_sweep16(h, Hp, bwd))
        launch_lstm_sweep16(st, bwd, Hp, grid, s);
    else if (bwd && h->bwd_mode == 1 && h->sweep_xchg) {
        LstmSweepArgs s8 = s;
        s8.bwd_waves = h->bwd_waves;
        // per-tile flags of the streamed hand-off: behind the XCD-id table of the same flag slice (zeroed with it)
        const bool whole = s.t0 == 0 && (s.t1 == 0 || s.t1 == s.T);   // time chunks (wavefront, pipelines) keep the flag forms
        s8.tflags = h->bwd_stream && whole ? s.xcc_tab + (size_t)((h->Bp + 7) / 8) * 64 : nullptr;
        s8.stash_via_lds |= h->bwd_dma << 3;
        if (h->bwd_stream == 2 && whole && h->sweep_xchg_tok && h->bwd_waves != 4 && s.xchg == h->sweep_xchg) {
            s8.token_handoff = h->token_early ? 1 : 2;   // 2: diagnostic, no early tile loads (PAULE_HIP_TOKEN_EARLY=0)
            s8.xchg = h->sweep_xchg_tok;
        }
        // prefetcher workgroups on the CUs the sweep leaves idle (lstm_persist_rs.hip: rs_prefetch_role)
        s8.n_pf = 0;
        if (s8.tflags && !s8.token_handoff && h->bwd_waves != 4 && !((s8.stash_via_lds >> 3) & 3)) {
            s8.n_pf = bwd_prefetchers_for(h, Hp, grid);
            s8.pf_dist = h->bwd_pf_dist;
        }
        // PAULE_HIP_BWD_CHAINS=C (round 5, probe): the chained form -- a workgroup serves C groups in turn, ceil(groups / C) sets on 8 slots
        if (h->bwd_chains > 0 && s8.tflags && !s8.token_handoff && h->bwd_waves != 4 && Hp == 736 && s.group_rows == 32 &&
            ((s.Bp + 31) / 32 + h->bwd_chains - 1) / h->bwd_chains <= 8 && h->n_cu >= 8 * (Hp / 32)) {
            s8.chains = h->bwd_chains;
            grid = 8 * (Hp / 32);
        }
        launch_lstm_bwd_rs_sweep(st, Hp, grid, s8);
    }
    else if (!bwd && h->dt == BF16 && h->sweep2 != 0 && s.group_rows == 32 && s.t0 == 0 && (s.t1 == 0 || s.t1 == s.T) &&
             lstm_fwd2_sweep_supported(Hp) && (h->sweep2 > 0 || (s.Bp + 31) / 32 > h->n_cu / (Hp / 32))) {
        // more 32-row groups than one pass of the one-per-CU forward sweep holds (cfg4's 2048 rows on one GPU: 64 groups in 6 passes of
        // 11): the two-per-CU recurrence role (lstm_fused2.hip) holds twice as many a pass at 1.4 x the step time
        const int P = Hp / 32, ng = (s.Bp + 31) / 32, fit = 2 * h->n_cu / P;
        // equal passes: 64 groups on 22 sets would be 22 + 22 + 20 -- the same three passes on 22 + 21 + 21 keep fewer workgroups per CU pair
        const int passes = (ng + fit - 1) / fit, sets = (ng + passes - 1) / passes;
        launch_lstm_fwd2_sweep(st, Hp, sets, s);
    }
    else
        launch_lstm_sweep(st, bwd, Hp, grid, s);
}

int sweep_per_xcd(pl_handle* h, const Model& md);
int wavefront_depth(pl_handle* h, const Model& md);
int wavefront_chunks(pl_handle* h, const Model& md, int Tl, int train_nb);
void model_forward_wavefront(pl_handle* h, hipStream_t st, Model& md, const void* in_act, int Tl, int nc, int lb, int le);
void model_backward_wavefront(pl_handle* h, hipStream_t st, Model& md, const void* dh_last, float* dIn, int Tl, int nc, int lb, int le);
// the predictor's first layer in a planning iteration: its streamed 32-row backward sweep carries dL/dCP as a ride-along tile (lstm_persist_rs.hip, XT)
bool pred_ride_along(pl_handle* h) {
    const int Hp = h->pred.Hp;
    return h->dx_part && h->dt == BF16 && !use_sweep16(h, Hp, true) && h->bwd_mode == 1 && h->sweep_xchg && h->bwd_stream == 1 && h->bwd_waves != 4 && h->bwd_dma == 0;
}
void set_ride_along(pl_handle* h, LstmLayer& ly, LstmSweepArgs& s) {
    s.WihT = ly.WihT;
    s.xpart = h->dx_part;
    s.skip_dA = h->bwd_xt == 2 ? 1 : 0;
}

// stacked LSTM forward over all Tl steps; in_act = time-major [Tl][Bp][in_p]
void model_forward(pl_handle* h, hipStream_t st, Model& md, const void* in_act, int Tl_use = 0) {
    const int Bp = h->Bp, Hp = md.Hp, Tl = Tl_use > 0 ? Tl_use : md.Tl;
    const size_t a = h->act;
    const int wf_nc = (h->wf_dirs & 1) ? wavefront_chunks(h, md, Tl, 0) : 0;
    const int wf_depth = wf_nc ? wavefront_depth(h, md) : 0;
    for (int l = 0; l < md.L; ++l) {
        if (wf_nc && md.L - l >= 2) {   // a band of layers as a wavefront
            const int le = l + wf_depth < md.L ? l + wf_depth : md.L;
            model_forward_wavefront(h, st, md, in_act, Tl, wf_nc, l, le);
            l = le - 1;
            continue;
        }
        const void* cur_in = l == 0 ? in_act : md.layers[l - 1].h;
        LstmLayer& ly = md.layers[l];
        const int sweep_grid = sweep_grid_for(h, Hp);
        // narrow inputs (CP, mel): the persistent sweep computes W_ih x_t + b itself; otherwise one batched GEMM for all
        // time steps: G = in * Wih^T + (b_ih + b_hh)
        const bool fuse_in = sweep_grid > 0 && h->fuse_input && (ly.in_p == 32 || ly.in_p == 64);
        if (!fuse_in)
            launch_gemm_nt(st, h->dt, false, cur_in, ly.in_p, ly.Wih, ly.in_p, ly.bias, ly.G, 4 * Hp, Tl * Bp, 4 * Hp, ly.in_p);
        if (sweep_grid > 0) {
            LstmSweepArgs s{};
            s.Bp = Bp;
            s.T = Tl;
            s.group_rows = sweep_group_rows_for(h, Hp);
            s.G = ly.G;
            s.W = ly.Whh;
            s.h = ly.h;
            s.c = ly.c;
            if (fuse_in) { s.x_in = cur_in; s.Wih = ly.Wih; s.bias = ly.bias; s.in_p = ly.in_p; }
            s.stash_via_lds = h->stash_lds ? 1 : 0;
            s.n_valid = (h->dt == F32 && h->f32_valu) ? h->rows_in_use : 0;
            s.counters = take_sweep_slice(h, st);
            s.flag_stride = h->flag_stride;
            s.xcc_tab = s.counters + (size_t)((Bp + 7) / 8) * h->T * h->flag_stride;
            // the forward same-XCD hand-off pays for the 16-row kernels (B = 8: 4.14 -> 4.02 ms, equal at B = 64), not for the 32-row ones
            s.xcd_fast = (h->xcd_fast & 1) | ((h->xcd_fast16 && (h->xcd_fast & 2) && use_sweep16(h, Hp, false)) ? 1 : 0);
            s.status = h->sweep_status;
            s.spin_ticks = h->spin_ticks; s.poll_mask = h->poll_mask;
            s.stamps = h->sweep_stamps;
            launch_sweep(h, st, false, Hp, sweep_grid, s);
            continue;
        }
        for (int t = 0; t < Tl; ++t) {
            LstmStepArgs s{};
            s.Bp = Bp;
            s.Hp = Hp;
            s.G_t = off(ly.G, (size_t)t * Bp * 4 * Hp, a);
            s.W = ly.Whh;
            s.h_prev = t ? off(ly.h, (size_t)(t - 1) * Bp * Hp, a) : nullptr;
            s.h_out = off(ly.h, (size_t)t * Bp * Hp, a);
            s.c_in = t ? h->c_run[(t - 1) & 1] : nullptr;
            s.c_out = h->c_run[t & 1];
            s.c_stash_t = off(ly.c, (size_t)t * Bp * Hp, a);
            launch_lstm_fwd_step(st, h->dt, s);
        }
    }
}

// backward-data through the stack.  Top-layer dL/dh is either dense (md.dh_ext, all steps) or, when
// dh_last != nullptr, a single [Bp][Hp] slab applied at the last step only (EmbeddingModel: only
// output[:, lens-1] feeds the loss, paule/models.py:442).  dIn: f32 time-major [Tl][Bp][in_p].
// train_nb > 0 (continued learning): the first train_nb batch rows carry samples; after each layer's recurrence the
// weight gradients dW_hh = sum_t dA_t^T h_{t-1}, dW_ih = sum_t dA_t^T in_t, db = sum_t dA_t are taken from the dA stash
// (in_act = the model's input slab), and the input gradient of layer 0 is not needed.
void model_backward(pl_handle* h, hipStream_t st, Model& md, const void* dh_last, float* dIn, int train_nb = 0,
                    const void* in_act = nullptr, int Tl_use = 0) {
    const int Bp = h->Bp, Hp = md.Hp, Tl = Tl_use > 0 ? Tl_use : md.Tl;
    const size_t a = h->act;
    const int wf_nc = (h->wf_dirs & 2) ? wavefront_chunks(h, md, Tl, train_nb) : 0;
    const int wf_depth = wf_nc ? wavefront_depth(h, md) : 0;
    for (int l = md.L - 1; l >= 0; --l) {
        if (wf_nc && l >= 1) {   // a band of layers as a wavefront
            const int lb = l + 1 - wf_depth > 0 ? l + 1 - wf_depth : 0;
            model_backward_wavefront(h, st, md, dh_last, dIn, Tl, wf_nc, lb, l + 1);
            l = lb;
            continue;
        }
        LstmLayer& ly = md.layers[l];
        const bool sparse_top = (l == md.L - 1) && dh_last;
        const int sweep_grid = sweep_grid_for(h, Hp, true);
        bool ride_along = false;
        if (sweep_grid > 0) {
            LstmSweepArgs s{};
            s.Bp = Bp;
            s.T = Tl;
            s.group_rows = sweep_group_rows_for(h, Hp, true);
            s.G = ly.G;
            s.W = ly.WhhT;
            s.c = ly.c;
            s.dh_ext = sparse_top ? nullptr : md.dh_ext;
            s.dh_last = sparse_top ? dh_last : nullptr;
            s.n_valid = (h->dt == F32 && h->f32_valu) ? h->rows_in_use : 0;
            s.counters = take_sweep_slice(h, st);
            s.flag_stride = h->flag_stride;
            s.xcc_tab = s.counters + (size_t)((Bp + 7) / 8) * h->T * h->flag_stride;
            s.xcd_fast = (h->xcd_fast >> 1) & 1;
            s.status = h->sweep_status;
            s.spin_ticks = h->spin_ticks; s.poll_mask = h->poll_mask;
            s.stamps = h->sweep_stamps ? h->sweep_stamps + 256 * 8 : nullptr;
            s.xchg = h->sweep_xchg;
            s.stash_via_lds = h->dt == F32 ? (h->wide_ingest ? 2 : 0) : ((h->own_store ? 2 : 0) | (h->wide_ingest16 ? 4 : 0));
            // the predictor's first layer in a planning iteration: dL/dCP = dA W_ih rides along in the sweep (partial tiles to dx_part,
            // summed by launch_dx_reduce below) where the streamed 32-row form runs the whole sequence; otherwise the batched product
            ride_along = l == 0 && &md == &h->pred && train_nb == 0 && dIn == h->dX && Tl == h->T && pred_ride_along(h);
            if (ride_along) { set_ride_along(h, ly, s); h->pred_dA_skipped = s.skip_dA != 0; }
            launch_sweep(h, st, true, Hp, sweep_grid, s);
        } else
        for (int t = Tl - 1; t >= 0; --t) {
            LstmStepArgs s{};
            s.Bp = Bp;
            s.Hp = Hp;
            s.G_t = off(ly.G, (size_t)t * Bp * 4 * Hp, a);
            s.G_next = (t + 1 < Tl) ? off(ly.G, (size_t)(t + 1) * Bp * 4 * Hp, a) : nullptr;
            s.W = ly.WhhT;
            s.c_in = (t + 1 < Tl) ? h->dc_run[(t + 1) & 1] : nullptr;
            s.c_out = h->dc_run[t & 1];
            s.c_stash_t = off(ly.c, (size_t)t * Bp * Hp, a);
            s.c_stash_prev = t ? off(ly.c, (size_t)(t - 1) * Bp * Hp, a) : nullptr;
            if (sparse_top)
                s.dh_ext = (t == Tl - 1) ? dh_last : nullptr;
            else
                s.dh_ext = off(md.dh_ext, (size_t)t * Bp * Hp, a);
            launch_lstm_bwd_step(st, h->dt, s);
        }
        if (train_nb > 0) {
            const void* lin = l == 0 ? in_act : md.layers[l - 1].h;
            launch_gemm_tn(st, h->dt, ly.G, 4 * Hp, ly.h, Hp, ly.gWhh, Hp, 4 * Hp, Hp, Bp, train_nb, Tl - 1, 1, 0, md.train_scratch,
                           md.train_scratch_bytes, h->n_cu, h->tn_bf16);
            launch_gemm_tn(st, h->dt, ly.G, 4 * Hp, lin, ly.in_p, ly.gWih, ly.in_p, 4 * Hp, ly.in_p, Bp, train_nb, Tl, 0, 0,
                           md.train_scratch, md.train_scratch_bytes, h->n_cu, h->tn_bf16);
            launch_colsum(st, h->dt, ly.G, 4 * Hp, 4 * Hp, Bp, train_nb, Tl, 0, ly.gb, md.colsum_part);
        }
        if (l > 0)   // dL/dh of the layer below = dA * Wih
            launch_gemm_nt(st, h->dt, false, ly.G, 4 * Hp, ly.WihT, 4 * Hp, nullptr, md.dh_ext, Hp, Tl * Bp, Hp, 4 * Hp);
        else if (train_nb > 0)
            ;
        else if (ride_along)
            launch_dx_reduce(st, h->dx_part, Hp, Bp, Tl, dIn);
        else
            launch_gemm_nt(st, h->dt, true, ly.G, 4 * Hp, ly.WihT, 4 * Hp, nullptr, dIn, ly.in_p, Tl * Bp, ly.in_p, 4 * Hp);
    }
}

// ---- layer wavefront -------------------------------------------------------------------------------------------
// A batch of one 16-row group occupies 23 (bf16) / 46 (f32) of the 256 CUs, and a stacked model runs its layers one after
// the other: L * T dependent steps.  Cut into time chunks the layers overlap -- layer l + 1 works on chunk c while layer l is
// on chunk c + 1 -- and the chain shrinks to about (chunks + L - 1) / chunks * T steps.  Every (layer, chunk) is one launch of
// the same sweep kernel on steps t0 .. t1-1 (flags and stashes of the chunk before are in place by stream order; the f32 cell
// state crosses the border through `carry`), the batched projections between the layers run per chunk, each layer has its own
// stream, events carry the (layer, chunk) dependencies, and the whole pattern is captured into the iteration's graph.
// All persistent workgroups of the sweeps that run side by side must be co-resident -- per XCD: workgroup b of a launch goes
// to XCD b % 8 and waits there for a free CU (these kernels take a CU each), so with two half-resident sweeps on one XCD
// neither would ever complete.  A band of at most depth = (CUs of an XCD) / (workgroups one sweep keeps on an XCD) layers
// forms one wavefront; a deeper model runs band after band.  H = 720 in bf16 pins a group's 23 workgroups to one XCD
// (the 8-slot grid): no second sweep fits beside it, and spreading the group instead costs more than the overlap gains.
// workgroups one sweep of the model keeps on an XCD (worst case); 0 = the model's sweeps cannot be cut into time chunks
int sweep_per_xcd(pl_handle* h, const Model& md) {
    const int Hp = md.Hp;
    if (md.L < 1 || sweep_grid_for(h, Hp, false) <= 0 || sweep_grid_for(h, Hp, true) <= 0) return 0;
    const int groups = (h->Bp + 15) / 16;
    if (h->dt == F32) {
        const int P = Hp / 16, grid = lstm_sweep_f32_grid(Hp, h->Bp, h->n_cu);
        if (grid < groups * P) return 0;   // the groups take turns already
        return (grid + 7) / 8;
    }
    if (use_sweep16(h, Hp, false) && use_sweep16(h, Hp, true)) {
        const int P = Hp / 32, slots = lstm_sweep16_grid(Hp, h->Bp, h->n_cu, h->small_grid) / P;
        if (slots < groups) return 0;
        return slots % 8 == 0 ? (groups + 7) / 8 * P : (slots * P + 7) / 8;   // slot s sits on XCD s % 8 / spread
    }
    return 0;
}

// Pipelines (not the layer wavefront) may launch the 16-row bf16 sweeps of fewer than 8 groups WITHOUT the 8 group slots: the
// group's workgroups then spread over all XCDs (ceil(groups * P / 8) per XCD whatever the rotation of the launch) instead of
// sitting together on one, which is what lets predictor and embedder sweeps of H = 720 run side by side.  The price is the
// same-XCD exchange, which chunked launches do not use anyway.
bool pipe_spread16(pl_handle* h, const Model& md) {
    if (h->dt != BF16 || !h->pipe_spread || !h->small_grid || md.L < 1) return false;
    if (!use_sweep16(h, md.Hp, false) || !use_sweep16(h, md.Hp, true)) return false;
    const int P = md.Hp / 32, groups = (h->Bp + 15) / 16;
    return groups < 8 && lstm_sweep16_grid(md.Hp, h->Bp, h->n_cu, false) == groups * P;
}
int pipe_per_xcd(pl_handle* h, const Model& md) {
    if (pipe_spread16(h, md)) return ((h->Bp + 15) / 16 * (md.Hp / 32) + 7) / 8;
    return sweep_per_xcd(h, md);
}
int pipe_grid(pl_handle* h, const Model& md, bool bwd) {
    if (pipe_spread16(h, md)) return lstm_sweep16_grid(md.Hp, h->Bp, h->n_cu, false);
    return sweep_grid_for(h, md.Hp, bwd);
}

int wavefront_depth(pl_handle* h, const Model& md) {
    const int per_xcd = sweep_per_xcd(h, md);
    if (md.L < 2 || per_xcd == 0) return 0;
    int depth = (h->n_cu / 8) / per_xcd;
    if (depth > md.L) depth = md.L;
    if (depth > 8) depth = 8;
    return depth >= 2 ? depth : 0;
}

// Cross-model pipeline of the acoustic path (small batches): predictor layers -> mel head + pooling -> embedder layers,
// chunk by chunk, when one sweep of each fits side by side on every XCD.  Returns the number of chunks (of mel frames) or 0.
int acoustic_pipeline_chunks(pl_handle* h) {
    if (h->wavefront <= 0 || !h->wf_pipeline || h->sweep_slot < 0 || !h->need_emb_in_step()) return 0;
    const Model &p = h->pred, &e = h->emb;
    if (e.L < 1 || p.L + e.L > 8 || h->emb_blocks > 0 || !p.layers[0].carry_f || !e.layers[0].carry_f) return 0;
    const int pp = pipe_per_xcd(h, p), pe = pipe_per_xcd(h, e);
    if (pp == 0 || pe == 0 || p.L * pp + e.L * pe > h->n_cu / 8) return 0;
    if (h->sweep_slot != 0 || 2 * (p.L + e.L) > h->n_layer_slots_zeroed) return 0;
    int nc = h->wavefront < 32 ? h->wavefront : 32;
    if (nc > h->Tp / 4) nc = h->Tp / 4;
    return nc >= 2 ? nc : 0;
}

int wavefront_chunks(pl_handle* h, const Model& md, int Tl, int train_nb) {
    if (h->wavefront <= 0 || h->sweep_slot < 0 || train_nb > 0 || !md.layers[0].carry_f) return 0;
    if (wavefront_depth(h, md) < 2) return 0;
    if (h->sweep_slot + md.L > h->n_layer_slots_zeroed) return 0;
    int nc = h->wavefront < 32 ? h->wavefront : 32;
    if (nc > Tl / 4) nc = Tl / 4;
    return nc >= 2 ? nc : 0;
}

// Layer streams and events live in one pool per device for the life of the process (callers hold the device's SweepChain
// mutex): hipStreamEndCapture crashed (segmentation fault inside the runtime, reproducible with
// tools/microbench/capture_stress.py) once streams that had taken part in an earlier capture had been destroyed with their
// handle and new ones were created for the next capture.  A handle only keeps cursors into the pool.
std::vector<hipStream_t>& wf_stream_pool(int dev) { static std::vector<hipStream_t> p[SweepChain::kMaxDev]; return p[dev]; }
std::vector<hipEvent_t>& wf_event_pool(int dev) { static std::vector<hipEvent_t> p[SweepChain::kMaxDev]; return p[dev]; }

hipStream_t wavefront_stream(pl_handle* h) {
    auto& pool = wf_stream_pool(h->cfg.device);
    if (h->wf_stream_next == pool.size()) {
        hipStream_t q = nullptr;
        (void)hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
        pool.push_back(q);
    }
    return pool[h->wf_stream_next++];
}

hipEvent_t wavefront_event(pl_handle* h) {
    auto& pool = wf_event_pool(h->cfg.device);
    if (h->wf_next == pool.size()) {
        hipEvent_t e = nullptr;
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        pool.push_back(e);
    }
    return pool[h->wf_next++];
}

// last node the capturing stream has recorded so far (null at the very start of the capture)
hipGraphNode_t capture_tail(hipStream_t st) {
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    const hipGraphNode_t* deps = nullptr;
    size_t n = 0;
    if (hipStreamGetCaptureInfo_v2(st, &status, nullptr, nullptr, &deps, &n) != hipSuccess || status != hipStreamCaptureStatusActive || n == 0)
        return nullptr;
    return deps[n - 1];
}

void fill_sweep_common(pl_handle* h, LstmSweepArgs& s, int Tl, int* slice) {
    s.Bp = h->Bp;
    s.T = Tl;
    s.group_rows = 16;
    s.counters = slice;
    s.flag_stride = h->flag_stride;
    s.xcc_tab = slice + (size_t)((h->Bp + 7) / 8) * h->T * h->flag_stride;
    s.xcd_fast = 0;   // the XCD table of a slice is filled once, by the first chunk's workgroups: later launches may sit elsewhere
    s.status = h->sweep_status;
    s.spin_ticks = h->spin_ticks; s.poll_mask = h->poll_mask;
    s.stamps = nullptr;
    s.n_valid = (h->dt == F32 && h->f32_valu) ? h->rows_in_use : 0;
}

// layers lb .. le-1 (le - lb <= 8) of the model as one wavefront of nc time chunks
void model_forward_wavefront(pl_handle* h, hipStream_t st, Model& md, const void* in_act, int Tl, int nc, int lb, int le) {
    const int Bp = h->Bp, Hp = md.Hp;
    const size_t a = h->act;
    const int grid = sweep_grid_for(h, Hp);
    int* slice[8];
    hipEvent_t ev[8] = {};   // ev[i]: layer lb + i finished its latest chunk
    hipStream_t ls[8];
    const bool cap = h->capturing;   // graph capture: everything on st, dependencies rewritten afterwards (build_graph)
    const int nl = le - lb;
    if (cap) h->wf_regions.emplace_back();
    for (int l = lb; l < le; ++l) {
        slice[l - lb] = take_sweep_slice(h, st);
        ls[l - lb] = (l == lb || cap) ? st : wavefront_stream(h);
    }
    for (int c = 0; c < nc; ++c) {
        const int t0 = (int)((long long)c * Tl / nc), t1 = (int)((long long)(c + 1) * Tl / nc);
        for (int l = lb; l < le; ++l) {
            LstmLayer& ly = md.layers[l];
            hipStream_t sl = ls[l - lb];
            pl_handle::WfSeg seg{cap ? capture_tail(st) : nullptr, nullptr, l > lb ? c * nl + (l - 1 - lb) : -1, c > 0 ? (c - 1) * nl + (l - lb) : -1};
            if (l > lb && !cap) (void)hipStreamWaitEvent(sl, ev[l - 1 - lb], 0);
            const void* cur_in = l == 0 ? in_act : md.layers[l - 1].h;
            const bool fuse_in = h->fuse_input && (ly.in_p == 32 || ly.in_p == 64);
            if (!fuse_in)
                launch_gemm_nt(sl, h->dt, false, off(const_cast<void*>(cur_in), (size_t)t0 * Bp * ly.in_p, a), ly.in_p, ly.Wih, ly.in_p,
                               ly.bias, off(ly.G, (size_t)t0 * Bp * 4 * Hp, a), 4 * Hp, (t1 - t0) * Bp, 4 * Hp, ly.in_p);
            LstmSweepArgs s{};
            fill_sweep_common(h, s, Tl, slice[l - lb]);
            s.G = ly.G; s.W = ly.Whh; s.h = ly.h; s.c = ly.c;
            if (fuse_in) { s.x_in = cur_in; s.Wih = ly.Wih; s.bias = ly.bias; s.in_p = ly.in_p; }
            s.stash_via_lds = h->stash_lds ? 1 : 0;
            s.t0 = t0; s.t1 = t1; s.carry = ly.carry_f;
            launch_sweep(h, sl, false, Hp, grid, s);
            if (cap) {
                seg.last = capture_tail(st);
                h->wf_regions.back().segs.push_back(seg);
                if (c == nc - 1) h->wf_regions.back().finals.push_back(c * nl + (l - lb));
            } else if (l < le - 1 || c == nc - 1) {
                ev[l - lb] = wavefront_event(h);
                (void)hipEventRecord(ev[l - lb], sl);
            }
        }
    }
    if (!cap)
        for (int l = lb + 1; l < le; ++l) (void)hipStreamWaitEvent(st, ev[l - lb], 0);   // join
}

void model_backward_wavefront(pl_handle* h, hipStream_t st, Model& md, const void* dh_last, float* dIn, int Tl, int nc, int lb, int le) {
    const int Bp = h->Bp, Hp = md.Hp, top = le - 1;
    const size_t a = h->act;
    const int grid = sweep_grid_for(h, Hp, true);
    int* slice[8];
    hipEvent_t ev[8] = {};   // ev[i]: layer lb + i finished its latest chunk, projection included
    hipStream_t ls[8];
    const bool cap = h->capturing;   // graph capture: everything on st, dependencies rewritten afterwards (build_graph)
    const int nl = le - lb;
    if (cap) h->wf_regions.emplace_back();
    for (int l = top; l >= lb; --l) {
        slice[l - lb] = take_sweep_slice(h, st);
        ls[l - lb] = (l == top || cap) ? st : wavefront_stream(h);
    }
    for (int c = nc - 1; c >= 0; --c) {
        const int t0 = (int)((long long)c * Tl / nc), t1 = (int)((long long)(c + 1) * Tl / nc);
        for (int l = top; l >= lb; --l) {
            LstmLayer& ly = md.layers[l];
            hipStream_t sl = ls[l - lb];
            // launch order: chunk nc-1 first, layer `top` first: segment (l, c) is number (nc-1-c) * nl + (top - l)
            pl_handle::WfSeg seg{cap ? capture_tail(st) : nullptr, nullptr, l < top ? (nc - 1 - c) * nl + (top - l - 1) : -1,
                                 c < nc - 1 ? (nc - 2 - c) * nl + (top - l) : -1};
            if (l < top && !cap) (void)hipStreamWaitEvent(sl, ev[l + 1 - lb], 0);
            const bool sparse_top = l == md.L - 1 && dh_last;
            LstmSweepArgs s{};
            fill_sweep_common(h, s, Tl, slice[l - lb]);
            s.G = ly.G; s.W = ly.WhhT; s.c = ly.c;
            s.dh_ext = sparse_top ? nullptr : md.dh_ext;
            s.dh_last = sparse_top ? dh_last : nullptr;
            s.xchg = ly.xchg;
            s.stash_via_lds = h->dt == F32 ? (h->wide_ingest ? 2 : 0) : ((h->own_store ? 2 : 0) | (h->wide_ingest16 ? 4 : 0));
            s.t0 = t0; s.t1 = t1; s.carry = ly.carry_b;
            launch_sweep(h, sl, true, Hp, grid, s);
            // dL/dh of the layer below (in place in md.dh_ext: the rows of chunk c are read by layer l before they are written
            // for layer l - 1, both in this stream's order), or the model's input gradient
            if (l > 0)
                launch_gemm_nt(sl, h->dt, false, off(ly.G, (size_t)t0 * Bp * 4 * Hp, a), 4 * Hp, ly.WihT, 4 * Hp, nullptr,
                               off(md.dh_ext, (size_t)t0 * Bp * Hp, a), Hp, (t1 - t0) * Bp, Hp, 4 * Hp);
            else
                launch_gemm_nt(sl, h->dt, true, off(ly.G, (size_t)t0 * Bp * 4 * Hp, a), 4 * Hp, ly.WihT, 4 * Hp, nullptr,
                               dIn + (size_t)t0 * Bp * ly.in_p, ly.in_p, (t1 - t0) * Bp, ly.in_p, 4 * Hp);
            if (cap) {
                seg.last = capture_tail(st);
                h->wf_regions.back().segs.push_back(seg);
                if (c == 0) h->wf_regions.back().finals.push_back((nc - 1 - c) * nl + (top - l));
            } else if (l > lb || c == 0) {
                ev[l - lb] = wavefront_event(h);
                (void)hipEventRecord(ev[l - lb], sl);
            }
        }
    }
    if (!cap)
        for (int l = lb; l < top; ++l) (void)hipStreamWaitEvent(st, ev[l - lb], 0);   // join
}

void pred_forward(pl_handle* h, hipStream_t st) {
    h->pred_dA_skipped = false;   // the forward pass rewrites the gate stash
    launch_pack_cp(st, h->dt, h->x, h->B, h->T, h->C, h->X0, h->Bp, h->Cp);
    model_forward(h, st, h->pred, h->X0);
    Model& p = h->pred;
    launch_gemm_nt(st, h->dt, true, p.layers[p.L - 1].h, p.Hp, p.Wlin, p.Hp, p.blin, h->Y, h->Mp, h->T * h->Bp, h->Mp, p.Hp);
    launch_pool_mel(st, h->dt, h->Y, h->B, h->T, h->M, h->Bp, h->Mp, h->mel_bm, h->mel_tm);
}

bool emb_ready(pl_handle* h) {
    if (!h->emb.ready()) return false;
    if (h->emb_post > 0 && !h->up_set) return false;
    for (char c : h->emb_conv_set)
        if (!c) return false;
    return true;
}

Model* model_by_id(pl_handle* h, int model_id) {
    switch (model_id) {
        case PL_MODEL_PRED: return &h->pred;
        case PL_MODEL_EMBED: return &h->emb;
        case PL_MODEL_INVERSE: return &h->inv;
        case PL_MODEL_CP_TUBE: return &h->tube;
        case PL_MODEL_TUBE_MEL: return &h->tmel;
        case PL_MODEL_TUBE_EMBED: return &h->temb;
        default: return nullptr;
    }
}

bool tube_ready(pl_handle* h) { return h->tube.ready() && h->tmel.ready() && h->temb.ready(); }

constexpr float kLeakySlope = 0.01f;   // torch.nn.LeakyReLU() default (paule/models.py:374, :425)

void emb_head_forward(pl_handle* h, hipStream_t st, const int32_t* lens);
// mel_bm: the embedder's input batch-major [B][Tp][M] f32 (read by the mel blocks only); h->mel_tm holds the same time-major
void emb_forward(pl_handle* h, hipStream_t st, const int32_t* lens, const float* mel_bm) {
    Model& e = h->emb;
    const void* in_tm = h->mel_tm;
    if (h->emb_blocks > 0) {   // x = x + MelChannelConv1D(x), emb_blocks times (paule/models.py:393-401)
        const size_t G = h->M / 3, blk = 3 * G * 16;
        const float* x = mel_bm;
        for (int i = 0; i < h->emb_blocks; ++i) {
            const float* w = h->emb_conv + i * blk;
            launch_mel_block(st, x, h->B, h->Tp, h->M, w, w + 3 * G * 15, h->emb_x[i & 1]);
            x = h->emb_x[i & 1];
        }
        launch_pack_mel(st, h->dt, x, h->B, h->Tp, h->M, h->emb_in, h->Bp, h->Mp);
        in_tm = h->emb_in;
    }
    model_forward(h, st, e, in_tm);
    emb_head_forward(h, st, lens);
}

// output at lens - 1, then linear_mapping or post_linear -> LeakyReLU -> output mapping
void emb_head_forward(pl_handle* h, hipStream_t st, const int32_t* lens) {
    Model& e = h->emb;
    launch_gather_last(st, h->dt, e.layers[e.L - 1].h, lens, h->B, e.Tl, h->Bp, e.Hp, h->h_last);
    if (h->emb_post == 0) {
        launch_gemm_nt(st, h->dt, true, h->h_last, e.Hp, e.Wlin, e.Hp, e.blin, h->sem, h->Sp, h->Bp, h->Sp, e.Hp);
        return;
    }
    // post_linear -> LeakyReLU -> output mapping (paule/models.py:405-407, :443-446)
    const int Pp = h->emb_post_p;
    launch_gemm_nt(st, h->dt, true, h->h_last, e.Hp, e.Wlin, e.Hp, e.blin, h->head_pre, Pp, h->Bp, Pp, e.Hp);
    launch_leaky(st, h->dt, h->head_pre, (int64_t)h->Bp * Pp, kLeakySlope, h->head_act);
    launch_gemm_nt(st, h->dt, true, h->head_act, Pp, h->Wup, Pp, h->bup, h->sem, h->Sp, h->Bp, h->Sp, Pp);
}

// dL/dsem (h->dsem) back through the embedder to dL/dmel (h->dmel_e, time-major f32 [Tp][Bp][Mp])
void emb_head_backward(pl_handle* h, hipStream_t st) {
    Model& e = h->emb;
    if (h->emb_post == 0) {
        // dL/dh_last = dsem * W_m
        launch_gemm_nt(st, h->dt, false, h->dsem, h->Sp, e.WlinT, h->Sp, nullptr, h->dv, e.Hp, h->Bp, e.Hp, h->Sp);
    } else {
        const int Pp = h->emb_post_p;
        launch_gemm_nt(st, h->dt, true, h->dsem, h->Sp, h->WupT, h->Sp, nullptr, h->head_d, Pp, h->Bp, Pp, h->Sp);
        launch_leaky_bwd(st, h->dt, h->head_d, h->head_pre, (int64_t)h->Bp * Pp, kLeakySlope, h->head_act);
        launch_gemm_nt(st, h->dt, false, h->head_act, Pp, e.WlinT, Pp, nullptr, h->dv, e.Hp, h->Bp, e.Hp, Pp);
    }
}

void emb_backward(pl_handle* h, hipStream_t st) {
    Model& e = h->emb;
    emb_head_backward(h, st);
    if (h->emb_blocks == 0) {
        model_backward(h, st, e, h->dv, h->dmel_e);
        return;
    }
    model_backward(h, st, e, h->dv, h->emb_din);
    // back through the residual mel blocks, last block first: time-major in, batch-major in between, time-major out
    const size_t G = h->M / 3, blk = 3 * G * 16;
    const int64_t sb_tm = h->Mp, st_tm = (int64_t)h->Bp * h->Mp, sb_bm = (int64_t)h->Tp * h->M, st_bm = h->M;
    const float* d = h->emb_din;
    bool d_tm = true;
    for (int i = h->emb_blocks - 1; i >= 0; --i) {
        const bool out_tm = i == 0;
        float* o = out_tm ? h->dmel_e : h->emb_x[i & 1];
        launch_mel_block_bwd(st, d, d_tm ? sb_tm : sb_bm, d_tm ? st_tm : st_bm, h->B, h->Tp, h->M, h->emb_conv + i * blk, o,
                             out_tm ? sb_tm : sb_bm, out_tm ? st_tm : st_bm);
        d = o;
        d_tm = out_tm;
    }
}

// somatosensory path, forward (paule/paule.py:916-919, :926-929): pred_tube = cp_tube_model(cp), pred_tube_mel =
// tube_mel_model(pred_tube) (pooled by two), pred_tube_semvec = tube_embedder(pred_tube, T)
void tube_heads_forward(pl_handle* h, hipStream_t st);
void tube_forward(pl_handle* h, hipStream_t st) {
    Model& u = h->tube;
    model_forward(h, st, u, h->X0);
    launch_gemm_nt(st, h->dt, false, u.layers[u.L - 1].h, u.Hp, u.Wlin, u.Hp, u.blin, h->tube_tm, h->Up, h->T * h->Bp, h->Up, u.Hp);
    tube_heads_forward(h, st);
}
// tube (h->tube_tm) -> pred_tube_mel (h->mel2_bm), pred_tube_semvec (h->sem2)
void tube_heads_forward(pl_handle* h, hipStream_t st) {
    Model &m = h->tmel, &e = h->temb;
    model_forward(h, st, m, h->tube_tm);
    launch_gemm_nt(st, h->dt, true, m.layers[m.L - 1].h, m.Hp, m.Wlin, m.Hp, m.blin, h->Y2, h->Mp, h->T * h->Bp, h->Mp, m.Hp);
    launch_pool_mel(st, h->dt, h->Y2, h->B, h->T, h->M, h->Bp, h->Mp, h->mel2_bm, h->mel2_tm);
    model_forward(h, st, e, h->tube_tm);
    launch_gather_last(st, h->dt, e.layers[e.L - 1].h, nullptr, h->B, e.Tl, h->Bp, e.Hp, h->h_last2);
    launch_gemm_nt(st, h->dt, true, h->h_last2, e.Hp, e.Wlin, e.Hp, e.blin, h->sem2, h->Sp, h->Bp, h->Sp, e.Hp);
}

// ... and backward: the two tube terms' gradients meet at pred_tube and go through the CP -> tube model into dX2
void tube_backward(pl_handle* h, hipStream_t st, const LossArgs& la) {
    Model &u = h->tube, &m = h->tmel, &e = h->temb;
    launch_dsem(st, h->dt, la, h->dsem2, true);
    launch_gemm_nt(st, h->dt, false, h->dsem2, h->Sp, e.WlinT, h->Sp, nullptr, h->dv2, e.Hp, h->Bp, e.Hp, h->Sp);
    model_backward(h, st, e, h->dv2, h->dtube_a);
    launch_dy(st, h->dt, la, nullptr, h->dY2, true);
    launch_gemm_nt(st, h->dt, false, h->dY2, h->Mp, m.WlinT, h->Mp, nullptr, m.dh_ext, m.Hp, h->T * h->Bp, m.Hp, h->Mp);
    model_backward(h, st, m, nullptr, h->dtube_b);
    launch_add2_act(st, h->dt, h->dtube_a, h->dtube_b, (int64_t)h->T * h->Bp * h->Up, h->dYt);
    launch_gemm_nt(st, h->dt, false, h->dYt, h->Up, u.WlinT, h->Up, nullptr, u.dh_ext, u.Hp, h->T * h->Bp, u.Hp, h->Up);
    model_backward(h, st, u, nullptr, h->dX2);
}

// ---- cross-model pipeline of the acoustic path (acoustic_pipeline_chunks) ---------------------------------------------------
// Stages = the predictor's layers (the top one also runs post_linear and the pooling of its frames), then the embedder's layers
// on the pooled frames of the same chunk.  Segment (stage, chunk) waits for (stage - 1, chunk) and (stage, chunk - 1): the structure of
// the layer wavefront with a model boundary inside, same eager / capture mechanics.
struct PipeCtx {
    pl_handle* h; hipStream_t st; bool cap; int nst;
    hipStream_t ls[8]; hipEvent_t ev[8];
    PipeCtx(pl_handle* hh, hipStream_t s, int n_stages) : h(hh), st(s), cap(hh->capturing), nst(n_stages) {
        if (cap) h->wf_regions.emplace_back();
        for (int i = 0; i < nst; ++i) { ls[i] = (i == 0 || cap) ? st : wavefront_stream(h); ev[i] = nullptr; }
    }
    pl_handle::WfSeg seg;
    hipStream_t begin(int stage, int k_chunk) {   // k_chunk: chunks in launch order
        seg = pl_handle::WfSeg{cap ? capture_tail(st) : nullptr, nullptr, stage > 0 ? k_chunk * nst + stage - 1 : -1,
                               k_chunk > 0 ? (k_chunk - 1) * nst + stage : -1};
        if (!cap && stage > 0) (void)hipStreamWaitEvent(ls[stage], ev[stage - 1], 0);
        return ls[stage];
    }
    void end(int stage, int k_chunk, bool last_chunk) {
        if (cap) {
            seg.last = capture_tail(st);
            h->wf_regions.back().segs.push_back(seg);
            if (last_chunk) h->wf_regions.back().finals.push_back(k_chunk * nst + stage);
        } else {
            ev[stage] = wavefront_event(h);
            (void)hipEventRecord(ev[stage], ls[stage]);
        }
    }
    void join() {
        if (!cap)
            for (int i = 1; i < nst; ++i) (void)hipStreamWaitEvent(st, ev[i], 0);
    }
};

void acoustic_forward_pipeline(pl_handle* h, hipStream_t st, int nc) {
    h->pred_dA_skipped = false;   // the forward pass rewrites the gate stash
    Model &p = h->pred, &e = h->emb;
    const int Bp = h->Bp, T = h->T, Tp = h->Tp;
    const size_t a = h->act;
    launch_pack_cp(st, h->dt, h->x, h->B, T, h->C, h->X0, Bp, h->Cp);
    int* slice[8];
    for (int l = 0; l < p.L + e.L; ++l) slice[l] = take_sweep_slice(h, st);
    PipeCtx px(h, st, p.L + e.L);   // stages: predictor layers 0 .. Lp-1, then embedder layers 0 .. Le-1
    const int grid_p = pipe_grid(h, p, false), grid_e = pipe_grid(h, e, false);
    for (int c = 0; c < nc; ++c) {
        const int e0 = (int)((long long)c * Tp / nc), e1 = (int)((long long)(c + 1) * Tp / nc);
        const int t0 = 2 * e0, t1 = c == nc - 1 ? T : 2 * e1;   // an odd last frame runs with the last chunk
        for (int l = 0; l < p.L; ++l) {   // predictor layers on frames t0 .. t1-1
            hipStream_t sl = px.begin(l, c);
            LstmLayer& ly = p.layers[l];
            const void* cur_in = l == 0 ? h->X0 : p.layers[l - 1].h;
            const bool fuse_in = h->fuse_input && (ly.in_p == 32 || ly.in_p == 64);
            if (!fuse_in)
                launch_gemm_nt(sl, h->dt, false, off(const_cast<void*>(cur_in), (size_t)t0 * Bp * ly.in_p, a), ly.in_p, ly.Wih, ly.in_p,
                               ly.bias, off(ly.G, (size_t)t0 * Bp * 4 * p.Hp, a), 4 * p.Hp, (t1 - t0) * Bp, 4 * p.Hp, ly.in_p);
            LstmSweepArgs s{};
            fill_sweep_common(h, s, T, slice[l]);
            s.G = ly.G; s.W = ly.Whh; s.h = ly.h; s.c = ly.c;
            if (fuse_in) { s.x_in = cur_in; s.Wih = ly.Wih; s.bias = ly.bias; s.in_p = ly.in_p; }
            s.stash_via_lds = h->stash_lds ? 1 : 0;
            s.t0 = t0; s.t1 = t1; s.carry = ly.carry_f;
            launch_sweep(h, sl, false, p.Hp, grid_p, s);
            px.end(l, c, c == nc - 1);
        }
        for (int l = 0; l < e.L; ++l) {   // embedder layers on the pooled frames e0 .. e1-1
            hipStream_t sl = px.begin(p.L + l, c);
            if (l == 0) {   // the mel head and the pooling of the chunk's frames run at the consumer: the predictor's next chunk
                            // (the long pole of the pipeline) does not queue behind them
                launch_gemm_nt(sl, h->dt, true, off(p.layers[p.L - 1].h, (size_t)t0 * Bp * p.Hp, a), p.Hp, p.Wlin, p.Hp, p.blin,
                               h->Y + (size_t)t0 * Bp * h->Mp, h->Mp, (t1 - t0) * Bp, h->Mp, p.Hp);
                launch_pool_mel(sl, h->dt, h->Y, h->B, T, h->M, Bp, h->Mp, h->mel_bm, h->mel_tm, e0, e1 - e0);
            }
            LstmLayer& ly = e.layers[l];
            const void* cur_in = l == 0 ? h->mel_tm : e.layers[l - 1].h;
            const bool fuse_in = h->fuse_input && (ly.in_p == 32 || ly.in_p == 64);
            if (!fuse_in)
                launch_gemm_nt(sl, h->dt, false, off(const_cast<void*>(cur_in), (size_t)e0 * Bp * ly.in_p, a), ly.in_p, ly.Wih, ly.in_p,
                               ly.bias, off(ly.G, (size_t)e0 * Bp * 4 * e.Hp, a), 4 * e.Hp, (e1 - e0) * Bp, 4 * e.Hp, ly.in_p);
            LstmSweepArgs s{};
            fill_sweep_common(h, s, Tp, slice[p.L + l]);
            s.G = ly.G; s.W = ly.Whh; s.h = ly.h; s.c = ly.c;
            if (fuse_in) { s.x_in = cur_in; s.Wih = ly.Wih; s.bias = ly.bias; s.in_p = ly.in_p; }
            s.stash_via_lds = h->stash_lds ? 1 : 0;
            s.t0 = e0; s.t1 = e1; s.carry = ly.carry_f;
            launch_sweep(h, sl, false, e.Hp, grid_e, s);
            px.end(p.L + l, c, c == nc - 1);
        }
    }
    px.join();
    emb_head_forward(h, st, nullptr);
}

// backward: embedder layers top .. 0, then the predictor layers top .. 0 (the top one starts with the dY rows of its frames and
// dY W_p), chunk by chunk from the last one
void acoustic_backward_pipeline(pl_handle* h, hipStream_t st, int nc, const LossArgs& la) {
    Model &p = h->pred, &e = h->emb;
    const int Bp = h->Bp, T = h->T, Tp = h->Tp;
    const size_t a = h->act;
    launch_dsem(st, h->dt, la, h->dsem);
    emb_head_backward(h, st);
    int* slice_e[8];
    int* slice_p[8];
    for (int l = e.L - 1; l >= 0; --l) slice_e[l] = take_sweep_slice(h, st);
    for (int l = p.L - 1; l >= 0; --l) slice_p[l] = take_sweep_slice(h, st);
    PipeCtx px(h, st, p.L + e.L);   // stage i < Le: embedder layer Le-1-i; stage Le + i: predictor layer Lp-1-i
    const int grid_p = pipe_grid(h, p, true), grid_e = pipe_grid(h, e, true);
    const int bwd_flags = h->dt == F32 ? (h->wide_ingest ? 2 : 0) : ((h->own_store ? 2 : 0) | (h->wide_ingest16 ? 4 : 0));
    for (int k = 0; k < nc; ++k) {
        const int c = nc - 1 - k;
        const int e0 = (int)((long long)c * Tp / nc), e1 = (int)((long long)(c + 1) * Tp / nc);
        const int t0 = 2 * e0, t1 = c == nc - 1 ? T : 2 * e1;
        for (int i = 0; i < e.L; ++i) {
            const int l = e.L - 1 - i;
            hipStream_t sl = px.begin(i, k);
            LstmLayer& ly = e.layers[l];
            const bool top = l == e.L - 1;
            LstmSweepArgs s{};
            fill_sweep_common(h, s, Tp, slice_e[l]);
            s.G = ly.G; s.W = ly.WhhT; s.c = ly.c;
            s.dh_ext = top ? nullptr : e.dh_ext;
            s.dh_last = top ? h->dv : nullptr;
            s.xchg = ly.xchg;
            s.stash_via_lds = bwd_flags;
            s.t0 = e0; s.t1 = e1; s.carry = ly.carry_b;
            launch_sweep(h, sl, true, e.Hp, grid_e, s);
            if (l > 0)
                launch_gemm_nt(sl, h->dt, false, off(ly.G, (size_t)e0 * Bp * 4 * e.Hp, a), 4 * e.Hp, ly.WihT, 4 * e.Hp, nullptr,
                               off(e.dh_ext, (size_t)e0 * Bp * e.Hp, a), e.Hp, (e1 - e0) * Bp, e.Hp, 4 * e.Hp);
            else {
                launch_gemm_nt(sl, h->dt, true, off(ly.G, (size_t)e0 * Bp * 4 * e.Hp, a), 4 * e.Hp, ly.WihT, 4 * e.Hp, nullptr,
                               h->dmel_e + (size_t)e0 * Bp * ly.in_p, ly.in_p, (e1 - e0) * Bp, ly.in_p, 4 * e.Hp);
                // dL/dY of the chunk's frames (it reads the input-gradient rows just written) and dL/dh_top = dY W_p of the predictor,
                // here at the producer: the predictor's stage (the long pole) is then the sweep alone
                launch_dy(sl, h->dt, la, h->dmel_e, h->dY, false, t0, t1 - t0);
                launch_gemm_nt(sl, h->dt, false, off(h->dY, (size_t)t0 * Bp * h->Mp, a), h->Mp, p.WlinT, h->Mp, nullptr,
                               off(p.dh_ext, (size_t)t0 * Bp * p.Hp, a), p.Hp, (t1 - t0) * Bp, p.Hp, h->Mp);
            }
            px.end(i, k, c == 0);
        }
        for (int i = 0; i < p.L; ++i) {
            const int l = p.L - 1 - i;
            hipStream_t sl = px.begin(e.L + i, k);
            LstmLayer& ly = p.layers[l];
            LstmSweepArgs s{};
            fill_sweep_common(h, s, T, slice_p[l]);
            s.G = ly.G; s.W = ly.WhhT; s.c = ly.c;
            s.dh_ext = p.dh_ext;
            s.xchg = ly.xchg;
            s.stash_via_lds = bwd_flags;
            s.t0 = t0; s.t1 = t1; s.carry = ly.carry_b;
            launch_sweep(h, sl, true, p.Hp, grid_p, s);
            if (l > 0)   // in place in p.dh_ext, as in the layer wavefront
                launch_gemm_nt(sl, h->dt, false, off(ly.G, (size_t)t0 * Bp * 4 * p.Hp, a), 4 * p.Hp, ly.WihT, 4 * p.Hp, nullptr,
                               off(p.dh_ext, (size_t)t0 * Bp * p.Hp, a), p.Hp, (t1 - t0) * Bp, p.Hp, 4 * p.Hp);
            px.end(e.L + i, k, c == 0);
        }
    }
    px.join();
    {   // dL/dCP = dA W_ih of the predictor's first layer: ONE product over all frames after the pipeline (nothing inside needs it,
        // and per chunk it would sit in the long pole: N = 32 with K = 4H is latency-bound whatever the number of rows)
        LstmLayer& ly = p.layers[0];
        launch_gemm_nt(st, h->dt, true, ly.G, 4 * p.Hp, ly.WihT, 4 * p.Hp, nullptr, h->dX, ly.in_p, T * Bp, ly.in_p, 4 * p.Hp);
    }
}

// one layer of a model on steps t0 .. t1-1 of its Tl-step recurrence (projection of the rows first unless it is fused)
void fwd_layer_chunk(pl_handle* h, hipStream_t sl, Model& md, int l, const void* in_act, int Tl, int t0, int t1, int* slice) {
    const int Bp = h->Bp, Hp = md.Hp;
    const size_t a = h->act;
    LstmLayer& ly = md.layers[l];
    const void* cur_in = l == 0 ? in_act : md.layers[l - 1].h;
    const bool fuse_in = h->fuse_input && (ly.in_p == 32 || ly.in_p == 64);
    if (!fuse_in)
        launch_gemm_nt(sl, h->dt, false, off(const_cast<void*>(cur_in), (size_t)t0 * Bp * ly.in_p, a), ly.in_p, ly.Wih, ly.in_p, ly.bias,
                       off(ly.G, (size_t)t0 * Bp * 4 * Hp, a), 4 * Hp, (t1 - t0) * Bp, 4 * Hp, ly.in_p);
    LstmSweepArgs s{};
    fill_sweep_common(h, s, Tl, slice);
    s.G = ly.G; s.W = ly.Whh; s.h = ly.h; s.c = ly.c;
    if (fuse_in) { s.x_in = cur_in; s.Wih = ly.Wih; s.bias = ly.bias; s.in_p = ly.in_p; }
    s.stash_via_lds = h->stash_lds ? 1 : 0;
    s.t0 = t0; s.t1 = t1; s.carry = ly.carry_f;
    launch_sweep(h, sl, false, Hp, pipe_grid(h, md, false), s);
}

// backward of one layer on steps t0 .. t1-1: the sweep, then dL/dh rows for the layer below (in place in md.dh_ext) or, for
// layer 0, the input gradient rows into dIn (f32 time-major [Tl][Bp][in_p]; null: the caller takes it whole afterwards)
void bwd_layer_chunk(pl_handle* h, hipStream_t sl, Model& md, int l, const void* dh_last, float* dIn, int Tl, int t0, int t1, int* slice) {
    const int Bp = h->Bp, Hp = md.Hp;
    const size_t a = h->act;
    LstmLayer& ly = md.layers[l];
    const bool sparse_top = l == md.L - 1 && dh_last;
    LstmSweepArgs s{};
    fill_sweep_common(h, s, Tl, slice);
    s.G = ly.G; s.W = ly.WhhT; s.c = ly.c;
    s.dh_ext = sparse_top ? nullptr : md.dh_ext;
    s.dh_last = sparse_top ? dh_last : nullptr;
    s.xchg = ly.xchg;
    s.stash_via_lds = h->dt == F32 ? (h->wide_ingest ? 2 : 0) : ((h->own_store ? 2 : 0) | (h->wide_ingest16 ? 4 : 0));
    s.t0 = t0; s.t1 = t1; s.carry = ly.carry_b;
    launch_sweep(h, sl, true, Hp, pipe_grid(h, md, true), s);
    if (l > 0)
        launch_gemm_nt(sl, h->dt, false, off(ly.G, (size_t)t0 * Bp * 4 * Hp, a), 4 * Hp, ly.WihT, 4 * Hp, nullptr,
                       off(md.dh_ext, (size_t)t0 * Bp * Hp, a), Hp, (t1 - t0) * Bp, Hp, 4 * Hp);
    else if (dIn)
        launch_gemm_nt(sl, h->dt, true, off(ly.G, (size_t)t0 * Bp * 4 * Hp, a), 4 * Hp, ly.WihT, 4 * Hp, nullptr,
                       dIn + (size_t)t0 * Bp * ly.in_p, ly.in_p, (t1 - t0) * Bp, ly.in_p, 4 * Hp);
}

// ---- the somatosensory path as a pipeline over time chunks (small batches) ------------------------------------------------------
// Stages: CP -> tube layers (the top one followed by its post_linear rows), the tube embedder's layers, the tube -> mel layers (the
// top one followed by post_linear rows and the pooling).  The chain order makes the tube -> mel stages wait for the embedder's
// stages of the same chunk too -- more than they need, but the pipeline keeps flowing and the bookkeeping stays a chain.
int tube_pipeline_chunks(pl_handle* h) {
    if (h->wavefront <= 0 || !h->wf_pipeline || h->sweep_slot < 0 || !h->tube_on()) return 0;
    const Model &u = h->tube, &m = h->tmel, &e = h->temb;
    if (u.L + m.L + e.L > 8) return 0;
    int per = 0;
    for (const Model* md : {&u, &m, &e}) {
        const int px = pipe_per_xcd(h, *md);
        if (px == 0 || !md->layers[0].carry_f) return 0;
        per += md->L * px;
    }
    if (per > h->n_cu / 8) return 0;
    if (h->sweep_slot + u.L + m.L + e.L > h->n_layer_slots_zeroed) return 0;
    int nc = h->wavefront < 32 ? h->wavefront : 32;
    if (nc > h->Tp / 4) nc = h->Tp / 4;
    return nc >= 2 ? nc : 0;
}

void tube_forward_pipeline(pl_handle* h, hipStream_t st, int nc) {
    Model &u = h->tube, &m = h->tmel, &e = h->temb;
    const int Bp = h->Bp, T = h->T, Tp = h->Tp, nst = u.L + e.L + m.L;
    const size_t a = h->act;
    int* slice[8];
    for (int i = 0; i < nst; ++i) slice[i] = take_sweep_slice(h, st);
    PipeCtx px(h, st, nst);
    for (int c = 0; c < nc; ++c) {
        const int e0 = (int)((long long)c * Tp / nc), e1 = (int)((long long)(c + 1) * Tp / nc);
        const int t0 = 2 * e0, t1 = c == nc - 1 ? T : 2 * e1;
        int stage = 0;
        for (int l = 0; l < u.L; ++l, ++stage) {
            hipStream_t sl = px.begin(stage, c);
            fwd_layer_chunk(h, sl, u, l, h->X0, T, t0, t1, slice[stage]);
            px.end(stage, c, c == nc - 1);
        }
        for (int l = 0; l < e.L; ++l, ++stage) {
            hipStream_t sl = px.begin(stage, c);
            if (l == 0)   // the tube's post_linear rows at the first consumer
                launch_gemm_nt(sl, h->dt, false, off(u.layers[u.L - 1].h, (size_t)t0 * Bp * u.Hp, a), u.Hp, u.Wlin, u.Hp, u.blin,
                               off(h->tube_tm, (size_t)t0 * Bp * h->Up, a), h->Up, (t1 - t0) * Bp, h->Up, u.Hp);
            fwd_layer_chunk(h, sl, e, l, h->tube_tm, T, t0, t1, slice[stage]);
            px.end(stage, c, c == nc - 1);
        }
        for (int l = 0; l < m.L; ++l, ++stage) {
            hipStream_t sl = px.begin(stage, c);
            fwd_layer_chunk(h, sl, m, l, h->tube_tm, T, t0, t1, slice[stage]);
            if (l == m.L - 1) {
                launch_gemm_nt(sl, h->dt, true, off(m.layers[l].h, (size_t)t0 * Bp * m.Hp, a), m.Hp, m.Wlin, m.Hp, m.blin,
                               h->Y2 + (size_t)t0 * Bp * h->Mp, h->Mp, (t1 - t0) * Bp, h->Mp, m.Hp);
                launch_pool_mel(sl, h->dt, h->Y2, h->B, T, h->M, Bp, h->Mp, h->mel2_bm, h->mel2_tm, e0, e1 - e0);
            }
            px.end(stage, c, c == nc - 1);
        }
    }
    px.join();
    launch_gather_last(st, h->dt, e.layers[e.L - 1].h, nullptr, h->B, e.Tl, Bp, e.Hp, h->h_last2);
    launch_gemm_nt(st, h->dt, true, h->h_last2, e.Hp, e.Wlin, e.Hp, e.blin, h->sem2, h->Sp, Bp, h->Sp, e.Hp);
}

void tube_backward_pipeline(pl_handle* h, hipStream_t st, int nc, const LossArgs& la) {
    Model &u = h->tube, &m = h->tmel, &e = h->temb;
    const int Bp = h->Bp, T = h->T, Tp = h->Tp, nst = u.L + e.L + m.L;
    const size_t a = h->act;
    launch_dsem(st, h->dt, la, h->dsem2, true);
    launch_gemm_nt(st, h->dt, false, h->dsem2, h->Sp, e.WlinT, h->Sp, nullptr, h->dv2, e.Hp, Bp, e.Hp, h->Sp);
    int* slice[8];
    for (int i = 0; i < nst; ++i) slice[i] = take_sweep_slice(h, st);
    PipeCtx px(h, st, nst);   // stages: embedder layers top .. 0, tube -> mel layers top .. 0, CP -> tube layers top .. 0
    for (int k = 0; k < nc; ++k) {
        const int c = nc - 1 - k;
        const int e0 = (int)((long long)c * Tp / nc), e1 = (int)((long long)(c + 1) * Tp / nc);
        const int t0 = 2 * e0, t1 = c == nc - 1 ? T : 2 * e1;
        int stage = 0;
        for (int l = e.L - 1; l >= 0; --l, ++stage) {
            hipStream_t sl = px.begin(stage, k);
            bwd_layer_chunk(h, sl, e, l, h->dv2, h->dtube_a, T, t0, t1, slice[stage]);
            px.end(stage, k, c == 0);
        }
        for (int l = m.L - 1; l >= 0; --l, ++stage) {
            hipStream_t sl = px.begin(stage, k);
            if (l == m.L - 1) {
                launch_dy(sl, h->dt, la, nullptr, h->dY2, true, t0, t1 - t0);
                launch_gemm_nt(sl, h->dt, false, off(h->dY2, (size_t)t0 * Bp * h->Mp, a), h->Mp, m.WlinT, h->Mp, nullptr,
                               off(m.dh_ext, (size_t)t0 * Bp * m.Hp, a), m.Hp, (t1 - t0) * Bp, m.Hp, h->Mp);
            }
            bwd_layer_chunk(h, sl, m, l, nullptr, h->dtube_b, T, t0, t1, slice[stage]);
            px.end(stage, k, c == 0);
        }
        for (int l = u.L - 1; l >= 0; --l, ++stage) {
            hipStream_t sl = px.begin(stage, k);
            if (l == u.L - 1) {   // the two gradient streams meet at the predicted tube
                launch_add2_act(sl, h->dt, h->dtube_a + (size_t)t0 * Bp * h->Up, h->dtube_b + (size_t)t0 * Bp * h->Up,
                                (int64_t)(t1 - t0) * Bp * h->Up, off(h->dYt, (size_t)t0 * Bp * h->Up, a));
                launch_gemm_nt(sl, h->dt, false, off(h->dYt, (size_t)t0 * Bp * h->Up, a), h->Up, u.WlinT, h->Up, nullptr,
                               off(u.dh_ext, (size_t)t0 * Bp * u.Hp, a), u.Hp, (t1 - t0) * Bp, u.Hp, h->Up);
            }
            bwd_layer_chunk(h, sl, u, l, nullptr, nullptr, T, t0, t1, slice[stage]);
            px.end(stage, k, c == 0);
        }
    }
    px.join();
    LstmLayer& l0 = u.layers[0];   // dL/dCP through the tube path, one product over all frames (as in the acoustic pipeline)
    launch_gemm_nt(st, h->dt, true, l0.G, 4 * u.Hp, l0.WihT, 4 * u.Hp, nullptr, h->dX2, l0.in_p, T * Bp, l0.in_p, 4 * u.Hp);
}

// ---- fused acoustic sweeps (lstm_fused.hip) ------------------------------------------------------------------------------
// Roles of the forward launch, in table order: 0 predictor recurrence, 1 mel head (post_linear + pooling), 2 embedder layer 1,
// then per further embedder layer l its input projection (3 + 2 (l - 1)) and its recurrence (4 + 2 (l - 1)).
// Placement (speed only): the grid has one block per CU, block b sits on XCD b % 8 (observed dealing); a recurrence set's
// P workgroups take consecutive depths of ONE XCD slot while one has room, everything else fills what is left.
struct FusedSet { int role, set, P; bool together; int npf = 0; };   // npf: stash prefetcher blocks wanted behind the set, on its slot (16-row backward recurrences)
// Role indices of the fused launches.  Forward: the predictor's layers first -- recurrence of layer l at 2 l, the projection that
// feeds it (l >= 1) at 2 l - 1 -- then the mel head, then the embedder: its first layer, and per further layer its projection and its
// recurrence.  With a one-layer predictor (the only shape the backward launch takes) that is 0 predictor, 1 head, 2 embedder layer 1,
// 3 + 2 (l - 1) / 4 + 2 (l - 1) projection / recurrence of embedder layer l: the numbering the backward roles use.
inline int fused_roles_count(int pred_layers, int emb_layers) { return (2 * pred_layers - 1) + 1 + (2 * emb_layers - 1); }
inline int fr_pred(int l) { return 2 * l; }
inline int fr_pred_proj(int l) { return 2 * l - 1; }
inline int fr_head(int pL) { return 2 * pL - 1; }
inline int fr_emb(int pL, int l) { return 2 * pL + 2 * l; }
inline int fr_emb_proj(int pL, int l) { return 2 * pL + 2 * l - 1; }

std::vector<short> fused_block_table(int n_cu, const std::vector<FusedSet>& sets, int* grid_out) {
    const int slots = 8, depth = n_cu / slots;
    std::vector<short> tab((size_t)slots * depth * 4, -1);
    std::vector<int> used(slots, 0), slot_of(sets.size(), -1);
    auto put = [&](int slot, const FusedSet& fs, int p, int pf) {
        const int b = used[slot]++ * slots + slot;
        tab[(size_t)b * 4] = (short)fs.role; tab[(size_t)b * 4 + 1] = (short)fs.set; tab[(size_t)b * 4 + 2] = (short)p; tab[(size_t)b * 4 + 3] = (short)pf;
    };
    for (int pass = 0; pass < 2; ++pass)
        for (size_t i = 0; i < sets.size(); ++i) {
            const FusedSet& fs = sets[i];
            if ((pass == 0) != fs.together) continue;
            int best = -1;
            if (fs.together)
                for (int s = 0; s < slots; ++s)
                    if (depth - used[s] >= fs.P && (best < 0 || used[s] < used[best])) best = s;
            if (best >= 0) {
                for (int p = 0; p < fs.P; ++p) put(best, fs, p, 0);
                slot_of[i] = best;
            } else {
                for (int p = 0; p < fs.P; ++p) {
                    int s = 0;
                    for (int k = 1; k < slots; ++k)
                        if (used[k] < used[s]) s = k;
                    if (used[s] >= depth) { *grid_out = 0; return tab; }
                    put(s, fs, p, 0);
                }
            }
        }
    // stash prefetchers (speed only) behind everything else: as many of a set's as its slot still holds; entry [3] = (count << 8) | (index + 1)
    for (size_t i = 0; i < sets.size(); ++i) {
        const FusedSet& fs = sets[i];
        if (fs.npf <= 0 || slot_of[i] < 0) continue;
        const int s = slot_of[i];
        const int n = fs.npf < depth - used[s] ? fs.npf : depth - used[s];
        for (int q = 0; q < n; ++q) put(s, fs, q, (n << 8) | (q + 1));
    }
    int top = 0;
    for (int s = 0; s < slots; ++s) top = used[s] > top ? used[s] : top;
    *grid_out = top * slots;
    return tab;
}

// chains per workgroup for the predictor's (Cp) and the embedder's (Ce) roles: the fewest CUs' worth of latency.  A
// chain-step keeps a workgroup busy ~2.1 us, a group's own step-to-step latency is ~4.2 us; the embedder has half the steps.
int plan_fused(pl_handle* h) {
    h->fused_fwd_ok = h->fused_bwd_ok = false;
    h->fused_fwd2 = false;
    // bit 0: fused forward launch, bit 1: fused backward launch.  Measured (profiles/r02_ab_fused_range.txt): up to 128 rows both
    // win (B = 64: 4.53 -> 3.33 ms per iteration, T = 2000: 30.4 -> 21.3; one or two chains per workgroup, the roles overlap and the
    // step latency hides); from 129 rows on the backward launch needs 2 - 4 chains per workgroup, its chain-step is bound by the
    // CU's memory pipe and the per-layer backward sweeps are faster (B = 256: 5.82 vs 7.19 ms) -- forward launch only.
    int mode = h->Bp >= 129 ? 1 : 3;
    // A STACKED predictor of another width than the embedder (the class-default 4 x 180 in front of 720: model set B) keeps both
    // launches at every batch size the roles fit: its recurrences are narrow (6 workgroups a set against 23), the chain counts stay
    // low, and the backward launch wins up to 256 rows as well (B = 64: 3.57 -> 2.49 ms per iteration, B = 256: 5.47 -> 4.52;
    // profiles/r03_ab_setB_fused_backward.txt).  PAULE_HIP_FUSED_BWD_STACKED=0: the backward stays on the per-layer sweeps.
    const bool two_width = !(h->pred.L == 1 && h->pred.Hp == h->emb.Hp);
    bool stacked32 = true;
    if (const char* z = std::getenv("PAULE_HIP_FUSED_BWD_STACKED")) stacked32 = std::atoi(z) != 0;
    bool mode_forced = false;
    if (two_width && stacked32) mode = 3;
    if (const char* z = std::getenv("PAULE_HIP_FUSED")) { mode = std::atoi(z); mode_forced = true; }
    int min_rows = 49;   // fewest rows on the 32-row roles; below: 16-row roles (rows16), else the chunk pipelines (4.1f) -- they beat the
                         // 32-row roles there (T = 2000, B = 32: 19.2 vs 20.8 ms)
    if (const char* z = std::getenv("PAULE_HIP_FUSED_MIN_B")) min_rows = std::atoi(z);
    // up to 48 rows (one to three 16-row groups: the reference's own B = 1, continued learning's 8, cfg5's 16 per GPU): both launches
    // with the LSTM roles on 16-row tiles and the same-XCD form of each role's own exchange (lstm_fused16.h) -- both or neither
    bool rows16 = h->Bp <= 48 && h->Bp < min_rows;   // one, two or three 16-row groups, each a set of its own in every LSTM role
    if (const char* z = std::getenv("PAULE_HIP_FUSED16")) rows16 = rows16 && std::atoi(z) != 0;
    if (rows16 && ((mode & 3) != 3 || !h->sweep16)) rows16 = false;   // PAULE_HIP_SWEEP16=0: no 16-row kernels of either kind
    const Model &p = h->pred, &e = h->emb;
    if (!(mode & 3) || h->dt != BF16 || !h->use_sweep || !h->fuse_input || (h->Bp < min_rows && !rows16)) return PL_OK;
    // the forward launch takes a stacked predictor (all its layers one width) in front of an embedder of another width (round 3: the
    // class-default 4 x 180 predictor of model set B); the backward launch one predictor layer and equal widths, as before
    const bool fwd_shape = p.L >= 1 && p.L <= 4 && e.L >= 1 && e.L <= 4 && fused_fwd_supported(p.Hp, e.Hp) && p.Hp / 32 <= 31 && e.Hp / 32 <= 31;
    // backward launch: one predictor layer and equal widths on 32-row tiles; on 16-row tiles (up to 16 rows) also the stacked
    // predictor of one width in front of an embedder of another (model set B)
    const bool bwd_shape = (rows16 || (stacked32 && two_width)) ? (p.L >= 1 && p.L <= 4 && e.L >= 1 && e.L <= 4 && fused_bwd16_supported(p.Hp, e.Hp))
                                                 : (p.L == 1 && p.Hp == e.Hp && fused_supported(p.Hp));
    if (!fwd_shape && !bwd_shape) return PL_OK;
    if (rows16 && (!fwd_shape || !bwd_shape || h->bwd_mode != 1)) return PL_OK;   // the chunk pipelines keep the shape
    if (!fwd_shape) mode &= ~1;
    if (!bwd_shape) mode &= ~2;
    if (!(mode & 3)) return PL_OK;
    if (p.layers[0].in_p != 32 || e.layers[0].in_p != 64 || h->Mp != 64 || h->emb_post > 0 || h->emb_blocks > 0) return PL_OK;
    if (!h->need_emb_in_step() || h->n_cu % 8 != 0) return PL_OK;
    const int Pp = p.Hp / 32, Pe = e.Hp / 32, ng = (h->Bp + 31) / 32, n_emb_roles = 2 * e.L - 1, n_pred_roles = 2 * p.L - 1;
    if (fused_roles_count(p.L, e.L) > kFusedMaxRoles) return PL_OK;
    int forced_p = 0, forced_e = 0;
    if (const char* z = std::getenv("PAULE_HIP_FUSED_CP")) forced_p = std::atoi(z);
    if (const char* z = std::getenv("PAULE_HIP_FUSED_CE")) forced_e = std::atoi(z);
    int best_cp = 0, best_ce = 0;
    const int ng16 = h->Bp / 16;
    const int ng_all = ng;
    if (rows16) {   // 16-row LSTM roles: one set per 16-row group, one chain; the product roles keep one set per 32-row group
        const int lstm_f = ng16 * (Pp * p.L + Pe * e.L), prod_f = ng * (Pp * (p.L - 1) + Pe * (e.L - 1) + 1);
        if (lstm_f + prod_f > h->n_cu) return PL_OK;   // forward and backward launch have the same counts
        best_cp = best_ce = 1;
    }
    bool occ2 = false;           // the forward launch at two workgroups per CU (lstm_fused2.hip): decided below
    int fwd_slots = h->n_cu;
    int ngf = ng;   // groups the role table is planned for
    // PAULE_HIP_FUSED_GPP=n: the forward launch in PASSES of n groups (lstm_fused.hip: fused_fwd_kernel walks every role over its sets pass
    // after pass) -- built in round 4 for batches of more groups than the roles hold at once (VERDICT r3 missing #5: cfg4's 2048 rows on one
    // GPU).  Bit-identical (test_fused_forward_in_passes_is_bit_identical), and NOT faster there: 43.9 ms per iteration in passes of 8 or
    // 16 groups against 41.4 ms on the per-layer forward sweeps, which at 64 groups already fill the chip (11 groups a pass on 253 CUs)
    // -- what the fused launch wins at 256 rows is the idle time of three half-empty sweeps (profiles/r04_ab_fused_passes.txt).  Opt-in.
    if (const char* z = fused_passes_compiled() ? std::getenv("PAULE_HIP_FUSED_GPP") : nullptr) {
        const int gf = std::atoi(z);
        if (gf > 0 && gf < ng && !rows16 && (mode & 1)) { ngf = gf; mode &= ~2; }
    }
    // chain counts for `slots` forward workgroup slots; o2: the two-per-CU forward launch (one or two chains a workgroup)
    auto search = [&](bool o2, int slots, int mode, int& bcp, int& bce) {   // mode: which launches the counts have to fit
        double bcost = 1e30;
        bcp = bce = 0;
        const int cmax = ((mode & 2) && h->bwd_mode == 1) ? 4 : kFusedMaxChains;   // the backward roles have LDS for 4 chains (lstm_fused.hip)
        for (int cp = 1; cp <= cmax; ++cp)
            for (int ce = 1; ce <= cmax; ++ce) {
                if (o2 && (cp > 2 || ce > 2)) continue;   // kFused2MaxChains (lstm_fused2.hip)
                if ((forced_p && cp != forced_p) || (forced_e && ce != forced_e)) continue;
                if (ngf != ng && (ngf % cp || ngf % ce)) continue;   // passes: whole sets only
                const int sp = (ngf + cp - 1) / cp, se = (ngf + ce - 1) / ce;
                // forward: the predictor's roles + one head workgroup per predictor set + the embedder's roles; backward: the
                // predictor's roles, the embedder's, one head workgroup per embedder set
                if ((mode & 1) && sp * (Pp * n_pred_roles + 1) + se * Pe * n_emb_roles > slots) continue;
                if ((mode & 2) && sp * Pp * n_pred_roles + se * (1 + Pe * n_emb_roles) > h->n_cu) continue;
                // two per CU: nothing of a chain-step hides behind the workgroup's other chain (the CU's other workgroup covers it): a
                // group's step takes the whole chain-step of every chain
                const double tp = o2 ? cp * 5.0 : std::max(cp * 2.1, 4.2), te = (o2 ? ce * 5.0 : std::max(ce * 2.1, 4.2)) / 2.0;
                const double cost = std::max(tp, te) + 1e-3 * (cp + ce);
                if (cost < bcost) { bcost = cost; bcp = cp; bce = ce; }
            }
    };
    for (int attempt = 0; attempt < 2 && !best_cp; ++attempt) {
        if (attempt == 1) {   // both launches do not fit the chip at this batch: the forward launch alone (unless the mode was asked for)
            if (mode_forced || rows16 || (mode & 3) != 3) break;
            mode &= ~2;
        }
        search(false, h->n_cu, mode, best_cp, best_ce);
    }
    int cp_b = best_cp, ce_b = best_ce;   // the backward launch keeps the shared counts (a one-per-CU forward launch too: counts of its own
                                          // changed nothing there -- set B at 64 ... 256 rows, set A at 64 / 128 were planned the same)
    // Two workgroups per CU for the forward launch (lstm_fused2.hip; PAULE_HIP_FUSED_OCC2 = 1 / 0 forces / forbids) -- with chain counts of
    // its own: the backward launch keeps the counts found above.  By itself only where it was measured to win
    // (profiles/r04_ab_fused_occ2.txt, interleaved A/B, ms per iteration at 300 frames):
    //  * a STACKED predictor of another width in front of the embedder (model set B: four narrow recurrences + three projection roles a
    //    group -- the roles that overlap worst on one wave per SIMD): every batch size of the 32-row roles -- 49 rows 2.48 -> 2.31, 64 2.49 ->
    //    2.31, 96 2.59 -> 2.37, 128 3.10 -> 2.92, 160 4.22 -> 3.38, 192 4.27 -> 3.50, 224 4.60 -> 3.84, 256 (cfg3_setB) 4.64 -> 4.17;
    //  * equal widths (set A): seven groups and more, and the predictor's role comes down to ONE chain per workgroup from more -- 224 /
    //    240 / 256 rows: 5.14 -> 5.01, 5.15 -> 5.01, 5.22 -> 5.04.  Not where the one-per-CU plan already has one or two chains: 128 rows
    //    3.51 -> 3.85, 160 4.57 -> 4.61, 192 4.65 -> 4.70, 128 x 2000 frames 22.6 -> 23.0.
    if (!rows16 && (mode & 1) && fwd_shape && fused_fwd2_supported(p.Hp, e.Hp) &&
        (h->fused_occ2 > 0 || (h->fused_occ2 < 0 && ngf == ng && (p.Hp < e.Hp || ng >= 7)))) {
        int cp2 = 0, ce2 = 0;
        search(true, 2 * h->n_cu, 1, cp2, ce2);
        if (cp2 && (h->fused_occ2 > 0 || p.Hp < e.Hp || (cp2 == 1 && best_cp > 1))) { occ2 = true; fwd_slots = 2 * h->n_cu; best_cp = cp2; best_ce = ce2; }
        if (occ2 && !cp_b) mode &= ~2;   // (forced, and nothing fitted one workgroup per CU: forward launch only)
    }
    if (!best_cp) return PL_OK;
    if (!cp_b) { cp_b = best_cp; ce_b = best_ce; }
    h->fused_gpp = ngf != ng_all ? ngf : 0;
    const int sp = (ngf + best_cp - 1) / best_cp, se = (ngf + best_ce - 1) / best_ce;
    const int sp_l = rows16 ? ng16 : sp, se_l = rows16 ? ng16 : se;   // sets of the LSTM roles
    h->fused_Cp = best_cp; h->fused_Ce = best_ce;
    h->fused_Cp_bwd = cp_b; h->fused_Ce_bwd = ce_b;
    const int sp_b = (ng + cp_b - 1) / cp_b, se_b = (ng + ce_b - 1) / ce_b;   // sets of the backward launch's roles
    const int sp_bl = rows16 ? ng16 : sp_b, se_bl = rows16 ? ng16 : se_b;
    h->fused_n_roles = fused_roles_count(p.L, e.L);
    int rc;
    if (mode & 1) {
        std::vector<FusedSet> sets;
        for (int l = 0; l < p.L; ++l)
            for (int s = 0; s < sp_l; ++s) sets.push_back({fr_pred(l), s, Pp, true});
        for (int s = 0; s < se_l; ++s) sets.push_back({fr_emb(p.L, 0), s, Pe, true});
        for (int l = 1; l < e.L; ++l)
            for (int s = 0; s < se_l; ++s) sets.push_back({fr_emb(p.L, l), s, Pe, true});
        for (int l = 1; l < p.L; ++l)
            for (int s = 0; s < sp; ++s) sets.push_back({fr_pred_proj(l), s, Pp, false});
        for (int l = 1; l < e.L; ++l)
            for (int s = 0; s < se; ++s) sets.push_back({fr_emb_proj(p.L, l), s, Pe, false});
        for (int s = 0; s < sp; ++s) sets.push_back({fr_head(p.L), s, 1, false});
        int grid = 0;
        std::vector<short> tab = fused_block_table(fwd_slots, sets, &grid);
        if (grid > 0 && grid <= fwd_slots) {
            h->fused_fwd2 = occ2;
            if (occ2 && h->fused2_xcd) {   // private tile-major copies of the recurrence roles' hand-off: [2 slots][groups][P][32 rows][64 B]
                for (int l = 0; l < p.L; ++l)
                    if ((rc = raw_alloc(h, &h->fused_hx[fr_pred(l)], (size_t)2 * ng * Pp * 2048))) return rc;
                for (int l = 0; l < e.L; ++l)
                    if ((rc = raw_alloc(h, &h->fused_hx[fr_emb(p.L, l)], (size_t)2 * ng * Pe * 2048))) return rc;
            }
            if ((rc = dev_alloc(h, &h->fused_tab_fwd, (size_t)fwd_slots * 4))) return rc;
            PL_HIP(hipMemcpyAsync(h->fused_tab_fwd, tab.data(), sizeof(short) * (size_t)grid * 4, hipMemcpyHostToDevice, h->stream));
            PL_HIP(hipStreamSynchronize(h->stream));   // tab is a local
            h->fused_grid_fwd = grid;
            h->fused_active_fwd = 0;
            for (int b = 0; b < grid; ++b) h->fused_active_fwd += tab[(size_t)b * 4] >= 0 ? 1 : 0;
            h->fused_fwd_ok = true;
        }
    }
    if ((mode & 2) && h->bwd_mode == 1) {
        // backward roles, numbered like the forward ones: fr_pred(l) the recurrence of predictor layer l, fr_pred_proj(l) (l >= 1) the
        // dL/dh product of layer l for the layer below, fr_head the backward mel head, fr_emb / fr_emb_proj the embedder's.  With one
        // predictor layer that is 0 predictor, 1 head, 2 embedder layer 1, 3 + 2 (l - 1) / 4 + 2 (l - 1) product / recurrence of layer l
        std::vector<FusedSet> sets;
        const int npf16 = rows16 && !two_width ? h->fused16_pf : 0;   // stash prefetchers behind every 16-row recurrence set of the equal-width launch (fused_pf_bwd16)
        for (int l = 0; l < p.L; ++l)
            for (int s = 0; s < sp_bl; ++s) sets.push_back({fr_pred(l), s, Pp, true, npf16});
        for (int l = 0; l < e.L; ++l)
            for (int s = 0; s < se_bl; ++s) sets.push_back({fr_emb(p.L, l), s, Pe, true, npf16});
        for (int l = 1; l < p.L; ++l)
            for (int s = 0; s < sp_b; ++s) sets.push_back({fr_pred_proj(l), s, Pp, false});
        for (int l = 1; l < e.L; ++l)
            for (int s = 0; s < se_b; ++s) sets.push_back({fr_emb_proj(p.L, l), s, Pe, false});
        for (int s = 0; s < se_b; ++s) sets.push_back({fr_head(p.L), s, 1, false});
        int grid = 0;
        std::vector<short> tab = fused_block_table(h->n_cu, sets, &grid);
        if (grid > 0 && grid <= h->n_cu) {
            if ((rc = dev_alloc(h, &h->fused_tab_bwd, (size_t)h->n_cu * 4))) return rc;
            PL_HIP(hipMemcpyAsync(h->fused_tab_bwd, tab.data(), sizeof(short) * (size_t)grid * 4, hipMemcpyHostToDevice, h->stream));
            PL_HIP(hipStreamSynchronize(h->stream));
            const size_t tile = 32 * 32 * 2;
            const size_t rec_p = 2 * (size_t)ng * Pp * Pp * tile, ring_p = (size_t)kFusedRing * ng * Pp * Pp * tile;
            const size_t rec_e = 2 * (size_t)ng * Pe * Pe * tile, ring_e = (size_t)kFusedRing * ng * Pe * Pe * tile;
            for (int l = 0; l < p.L; ++l) {
                if ((rc = raw_alloc(h, &h->fused_xchg[fr_pred(l)], rec_p))) return rc;
                if (l >= 1 && (rc = raw_alloc(h, &h->fused_xchg_ext[fr_pred_proj(l)], ring_p))) return rc;
                if (l >= 1 && (rc = alloc_act(h, &h->fused_dh_pred[l - 1], (size_t)h->T * h->Bp * p.Hp))) return rc;
            }
            for (int l = 0; l < e.L; ++l) {
                if ((rc = raw_alloc(h, &h->fused_xchg[fr_emb(p.L, l)], rec_e))) return rc;
                if (l >= 1 && (rc = raw_alloc(h, &h->fused_xchg_ext[fr_emb_proj(p.L, l)], ring_e))) return rc;
            }
            if ((rc = raw_alloc(h, &h->fused_xchg_mel, (size_t)kFusedRing * ng * (h->Mp / 32) * Pe * tile))) return rc;
            h->fused_grid_bwd = grid;
            h->fused_active_bwd = 0;
            for (int b = 0; b < grid; ++b) h->fused_active_bwd += tab[(size_t)b * 4] >= 0 ? 1 : 0;
            h->fused_bwd_ok = true;
        }
    }
    if (rows16) {
        if (h->fused_fwd_ok && h->fused_bwd_ok) {
            h->fused_rows16 = true;
            for (int l = 0; l < p.L; ++l)
                if ((rc = raw_alloc(h, &h->fused_hx[fr_pred(l)], (size_t)2 * ng16 * 16 * p.Hp * 2))) return rc;
            for (int l = 0; l < e.L; ++l)
                if ((rc = raw_alloc(h, &h->fused_hx[fr_emb(p.L, l)], (size_t)2 * ng16 * 16 * e.Hp * 2))) return rc;
        } else {
            h->fused_fwd_ok = h->fused_bwd_ok = false;
        }
    }
    return PL_OK;
}

int* fused_slice(pl_handle* h, int r, bool bwd);
void fused_common_args(pl_handle* h, FusedArgs& a, int grid, const short* tab, const FusedRole* roles, bool bwd) {
    a.Bp = h->Bp; a.B = h->B; a.n_groups = (h->Bp + 31) / 32; a.flag_stride = h->flag_stride; a.n_roles = h->fused_n_roles;
    a.grid = grid;
    a.gpp = bwd ? 0 : h->fused_gpp;
    a.status = h->sweep_status; a.spin_ticks = h->spin_ticks; a.poll_mask = h->poll_mask;
    a.block_tab = tab;
    a.roles = roles;
    // the last word of role 0's flag slice (inside its XCD-id table, which the fused launches do not use): zeroed with the flags
    a.census = fused_slice(h, 0, bwd) + h->sweep_cnt_bytes / sizeof(int) - 1;
    a.n_active = bwd ? h->fused_active_bwd : h->fused_active_fwd;
    if (h->census_hooks) {   // test hooks: a workgroup that never shows up / one that shows up after the others have given up
        long long z = 0;
        if (debug_value("census_expect_extra", &z)) a.n_active += (int)z;
        if (debug_value("census_late_ms", &z)) a.census_late_ticks = 100000ull * (unsigned long long)z;
    }
    a.prio = bwd ? 0 : (h->fused2_prio >= 0 ? h->fused2_prio : ((h->pred.L == 1 && h->fused_Cp == 1 && h->fused_Ce >= 2) ? 1 : 0));
    a.census_ticks = h->census_ticks;
    if (h->debug_fused)
        fprintf(stderr, "[pl] fused %s launch: grid %d, %d role-bearing workgroups expected, census word at slice int %zu, bound %llu ticks, Cp %d Ce %d\n",
                bwd ? "backward" : "forward", grid, a.n_active, (size_t)(a.census - h->sweep_cnt), a.census_ticks, h->fused_Cp, h->fused_Ce);
    a.stamps = h->sweep_stamps ? h->sweep_stamps + (bwd ? 256 * 8 : 0) : nullptr;
}

// flag slice of role r of the fused forward (bwd = false) / backward launch: the last 2 n_roles slices of an iteration
int* fused_slice(pl_handle* h, int r, bool bwd) {   // backward: r >= n_roles are the second flag sets of the product roles
    const size_t ints = h->sweep_cnt_bytes / sizeof(int);
    const int n_fused = 2 * h->fused_n_roles + (h->emb.L - 1) + (h->pred.L - 1);
    return h->sweep_cnt + (size_t)(h->n_sweep_slots - n_fused + (bwd ? h->fused_n_roles : 0) + r) * ints;
}

// 16-row LSTM roles (fused_rows16): the second, plain flag set and the XCD-id table of the role sit in its own flag slice -- a slice
// has room for (Bp + 7) / 8 = 2 groups of arrival flags and these launches have one: the second group's area holds the plain set,
// the XCD ids follow where they do for the per-layer sweeps
// 32-row backward LSTM roles (round 4): the XCD-id table only -- a verified same-XCD set hands its own partial tiles over through the
// shared L2 (plain stores, nt LDS-DMA; flags unchanged).  One row of 64 per SET of the role, where the per-layer sweeps keep one per group.
// PAULE_HIP_XCD_FAST bit 1 = 0 or PAULE_HIP_FUSED_XCD=0: write-through everywhere (A/B, counter passes)
void fused32_xcd_fields(pl_handle* h, FusedRole& R, int* slice) {
    if (h->fused_rows16 || !(h->xcd_fast & 2) || !h->fused_xcd) return;
    // three and more chains per workgroup hide the hand-off behind the other chains' work: nothing to gain there (set B at 256 rows: 4.66
    // against 4.68 ms per iteration), so the role keeps the one write-through form (profiles/r04_ab_fused_bwd_xcd.txt)
    if (R.C > 2) return;
    R.xtab = slice + (size_t)((h->Bp + 7) / 8) * h->T * h->flag_stride;
    // the plain flag set of the own exchange: the second quarter of the slice (the 32-row groups' write-through flags fill the first)
    R.fast_flags = slice + (size_t)((h->Bp + 31) / 32) * h->T * h->flag_stride;
}

// two-per-CU forward recurrence roles (lstm_fused2.hip, round 5): XCD-id table (one row of 64 per set), the plain flag set of the own exchange
// (second quarter of the slice, as for the 32-row backward roles) and the private tile-major copy of the hand-off
void fused2_fwd_xcd_fields(pl_handle* h, FusedRole& R, int* slice, void* hx) {
    if (h->fused_rows16 || !h->fused_fwd2 || !h->fused2_xcd || !hx) return;
    R.xtab = slice + (size_t)((h->Bp + 7) / 8) * h->T * h->flag_stride;
    R.fast_flags = slice + (size_t)((h->Bp + 31) / 32) * h->T * h->flag_stride;
    R.hx = hx;
}

void fused16_fields(pl_handle* h, FusedRole& R, int* slice, void* hx) {
    if (!h->fused_rows16) return;
    if (!h->xcd_fast || !h->xcd_fast16) return;   // PAULE_HIP_XCD_FAST=0 / PAULE_HIP_XCD_FAST16=0: the write-through exchange everywhere (A/B, counter passes)
    // the slice holds (Bp + 7) / 8 = 2 x (Bp / 16) groups' worth of arrival flags: the write-through set of the 16-row groups first, the
    // plain set behind it; one XCD-id table of 64 per group follows, as for the per-layer sweeps
    R.fast_flags = slice + (size_t)(h->Bp / 16) * h->T * h->flag_stride;
    R.xtab = slice + (size_t)((h->Bp + 7) / 8) * h->T * h->flag_stride;
    R.hx = hx;
}

// the role tables (called once, at the end of pl_create: every buffer exists)
int build_fused_roles(pl_handle* h) {
    Model &p = h->pred, &e = h->emb;
    const int n_roles = h->fused_n_roles, T = h->T, Tp = h->Tp;
    int rc;
    if (h->fused_fwd_ok) {
        std::vector<FusedRole> roles(n_roles);
        int* fl[kFusedMaxRoles];
        for (int r = 0; r < n_roles; ++r) fl[r] = fused_slice(h, r, false);
        const int Pp = p.Hp / 32, Pe = e.Hp / 32, pL = p.L;
        for (int l = 0; l < pL; ++l) {   // the predictor's layers: CP input fused into the first, a projection role in front of every further one
            LstmLayer& ly = p.layers[l];
            FusedRole& R = roles[fr_pred(l)];
            R.type = FR_LSTM_FWD; R.wide = 0; R.C = h->fused_Cp; R.T = T; R.flags = fl[fr_pred(l)];
            R.wait[0] = FusedWait{fl[fr_pred(l)], T, Pp, 0, 0, -1};
            R.G = ly.G; R.W = ly.Whh; R.h = ly.h; R.c = ly.c;
            fused16_fields(h, R, fl[fr_pred(l)], h->fused_hx[fr_pred(l)]);
            fused2_fwd_xcd_fields(h, R, fl[fr_pred(l)], h->fused_hx[fr_pred(l)]);
            if (l == 0) {
                R.ksx = ly.in_p / 16; R.x_in = h->X0; R.Wih = ly.Wih; R.bias = ly.bias;
            } else {
                R.ksx = 0; R.src_sc1 = 1;
                R.wait[2] = FusedWait{fl[fr_pred_proj(l)], T, 1, 1, 0, 0};
                FusedRole& Rp = roles[fr_pred_proj(l)];
                Rp.type = FR_PROJ_FWD; Rp.wide = 0; Rp.C = h->fused_Cp; Rp.T = T; Rp.flags = fl[fr_pred_proj(l)];
                Rp.wait[0] = FusedWait{fl[fr_pred(l - 1)], T, Pp, 0, 0, 0};
                Rp.src_h = p.layers[l - 1].h; Rp.Wg = ly.Wih; Rp.bias = ly.bias; Rp.out = ly.G;
            }
        }
        {   // mel head on every step of the predictor's top layer, pooled pairs out
            FusedRole& R = roles[fr_head(pL)];
            R.type = FR_HEAD_FWD; R.wide = 0; R.C = h->fused_Cp; R.T = 2 * Tp; R.flags = fl[fr_head(pL)];
            R.wait[0] = FusedWait{fl[fr_pred(pL - 1)], T, Pp, 0, 0, 0};
            R.src_h = p.layers[pL - 1].h; R.Wg = p.Wlin; R.bias = p.blin; R.out = h->mel_tm; R.out_bm = h->mel_bm; R.out_dim = h->M; R.out_p = h->Mp;
        }
        {   // embedder layer 1, pooled mel input fused
            FusedRole& R = roles[fr_emb(pL, 0)];
            LstmLayer& ly = e.layers[0];
            R.type = FR_LSTM_FWD; R.wide = 1; R.ksx = ly.in_p / 16; R.C = h->fused_Ce; R.T = Tp; R.flags = fl[fr_emb(pL, 0)];
            R.wait[0] = FusedWait{fl[fr_emb(pL, 0)], Tp, Pe, 0, 0, -1};
            R.wait[1] = FusedWait{fl[fr_head(pL)], Tp, 1, 0, 0, 0};
            R.src_sc1 = 1;
            R.G = ly.G; R.W = ly.Whh; R.h = ly.h; R.c = ly.c; R.x_in = h->mel_tm; R.Wih = ly.Wih; R.bias = ly.bias;
            fused16_fields(h, R, fl[fr_emb(pL, 0)], h->fused_hx[fr_emb(pL, 0)]);
            fused2_fwd_xcd_fields(h, R, fl[fr_emb(pL, 0)], h->fused_hx[fr_emb(pL, 0)]);
        }
        for (int l = 1; l < e.L; ++l) {
            LstmLayer& ly = e.layers[l];
            const int rp = fr_emb_proj(pL, l), rl = fr_emb(pL, l), rsrc = fr_emb(pL, l - 1);
            FusedRole& Rp = roles[rp];
            Rp.type = FR_PROJ_FWD; Rp.wide = 1; Rp.C = h->fused_Ce; Rp.T = Tp; Rp.flags = fl[rp];
            Rp.wait[0] = FusedWait{fl[rsrc], Tp, Pe, 0, 0, 0};
            Rp.src_h = e.layers[l - 1].h; Rp.Wg = ly.Wih; Rp.bias = ly.bias; Rp.out = ly.G;
            FusedRole& Rl = roles[rl];
            Rl.type = FR_LSTM_FWD; Rl.wide = 1; Rl.ksx = 0; Rl.C = h->fused_Ce; Rl.T = Tp; Rl.flags = fl[rl];
            Rl.wait[0] = FusedWait{fl[rl], Tp, Pe, 0, 0, -1};
            Rl.wait[2] = FusedWait{fl[rp], Tp, 1, 1, 0, 0};
            Rl.src_sc1 = 1;
            Rl.G = ly.G; Rl.W = ly.Whh; Rl.h = ly.h; Rl.c = ly.c;
            fused16_fields(h, Rl, fl[rl], h->fused_hx[rl]);
            fused2_fwd_xcd_fields(h, Rl, fl[rl], h->fused_hx[rl]);
        }
        if ((rc = dev_alloc(h, &h->fused_roles_fwd, (size_t)n_roles))) return rc;
        PL_HIP(hipMemcpyAsync(h->fused_roles_fwd, roles.data(), sizeof(FusedRole) * n_roles, hipMemcpyHostToDevice, h->stream));
        PL_HIP(hipStreamSynchronize(h->stream));
    }
    if (h->fused_bwd_ok) {
        std::vector<FusedRole> roles(n_roles);
        int* fl[kFusedMaxRoles];
        for (int r = 0; r < n_roles; ++r) fl[r] = fused_slice(h, r, true);
        const int Pp = p.Hp / 32, Pe = e.Hp / 32, pL = p.L;
        const int r_head = fr_head(pL), r_emb0 = fr_emb(pL, 0);
        // second flag sets ("the reduced dL/dh rows of a step are in place") of the product roles: the embedder's first, then the predictor's
        auto flags2_emb = [&](int l) { return fused_slice(h, n_roles + (l - 1), true); };               // product role of embedder layer l >= 1
        auto flags2_pred = [&](int l) { return fused_slice(h, n_roles + (e.L - 1) + (l - 1), true); };   // ... of predictor layer l >= 1
        for (int l = 0; l < pL; ++l) {   // the predictor's recurrences: the top layer takes dL/dh from the backward mel head (one row per pooled
            LstmLayer& ly = p.layers[l];                                   // frame), every lower layer from the product role of the layer above
            const int rl = fr_pred(l);
            const bool top = l == pL - 1;
            FusedRole& R = roles[rl];
            R.type = FR_LSTM_BWD; R.wide = 0; R.C = h->fused_Cp_bwd; R.T = T; R.flags = fl[rl];
            R.wait[0] = FusedWait{fl[rl], T, Pp, 0, 0, 1};
            R.src_sc1 = 1;
            R.G = ly.G; R.W = ly.WhhT; R.c = ly.c; R.xchg = h->fused_xchg[rl];
            if (top) {
                R.wait[2] = FusedWait{fl[r_head], Tp, 1, 0, 1, 0};
                R.dh_ext = p.dh_ext; R.dh_ext_half = 1; R.dh_ext_rows = Tp;
            } else {
                R.wait[1] = FusedWait{flags2_pred(l + 1), T, 1, 1, 0, 0};
                R.dh_ext = h->fused_dh_pred[l]; R.dh_ext_half = 0; R.dh_ext_rows = T;
            }
            fused16_fields(h, R, fl[rl], nullptr);
            fused32_xcd_fields(h, R, fl[rl]);
            if (l >= 1) {   // this layer's dA feeds its product role: dL/dh of the layer below
                R.dA_sc1 = 1;
                const int rdx = fr_pred_proj(l);
                FusedRole& D = roles[rdx];
                D.type = FR_DX_BWD; D.wide = 0; D.C = h->fused_Cp_bwd; D.T = T; D.flags = fl[rdx];
                D.flags2 = flags2_pred(l);
                D.wait[0] = FusedWait{D.flags2, T, Pp, 0, 0, kFusedRing};
                D.wait[2] = FusedWait{fl[rl], T, 1, 1, 0, 0};
                D.G = ly.G; D.Wg = ly.WihT; D.xchg_ext = h->fused_xchg_ext[rdx]; D.out = h->fused_dh_pred[l - 1];
            }
        }
        {   // backward mel head: input-gradient tiles of the embedder's first layer in, dL/dh rows of the predictor's top layer out
            FusedRole& R = roles[r_head];
            R.type = FR_HEAD_BWD; R.wide = 0; R.C = h->fused_Ce_bwd; R.T = Tp; R.flags = fl[r_head];
            R.wait[0] = FusedWait{fl[r_emb0], Tp, Pe, 0, 0, 0};
            R.Wg = p.WlinT; R.out = p.dh_ext; R.dh_ext = h->Y; R.out_dim = h->M; R.out_p = h->Mp; R.xchg_mel = h->fused_xchg_mel;
        }
        for (int l = 0; l < e.L; ++l) {
            LstmLayer& ly = e.layers[l];
            const int rl = fr_emb(pL, l);
            const bool top = l == e.L - 1;
            FusedRole& R = roles[rl];
            R.type = FR_LSTM_BWD; R.wide = 1; R.C = h->fused_Ce_bwd; R.T = Tp; R.flags = fl[rl];
            R.wait[0] = FusedWait{fl[rl], Tp, Pe, 0, 0, 1};
            R.G = ly.G; R.W = ly.WhhT; R.c = ly.c; R.xchg = h->fused_xchg[rl];
            fused16_fields(h, R, fl[rl], nullptr);
            fused32_xcd_fields(h, R, fl[rl]);
            if (top) R.dh_last = h->dv;
            else {   // dL/dh rows from the layer above's product role (reduced there): this slice's columns come from its slice-p workgroup
                R.wait[1] = FusedWait{flags2_emb(l + 1), Tp, 1, 1, 0, 0};
                R.dh_ext = e.dh_ext; R.dh_ext_rows = Tp; R.src_sc1 = 1;
            }
            if (l == 0) {   // input-gradient tiles for the backward mel head; a ring slot is free once the head has finished the step that used it
                R.wait[2] = FusedWait{fl[r_head], Tp, 1, 0, 0, kFusedRing};
                R.Wg = ly.WihT; R.out_p = h->Mp; R.xchg_mel = h->fused_xchg_mel;
            } else {        // this layer's dA feeds its product role
                R.dA_sc1 = 1;
                const int rdx = fr_emb_proj(pL, l);
                FusedRole& D = roles[rdx];
                D.type = FR_DX_BWD; D.wide = 1; D.C = h->fused_Ce_bwd; D.T = Tp; D.flags = fl[rdx];
                D.flags2 = flags2_emb(l);
                // a ring slot of partial tiles is free once every workgroup of the set has reduced the step that used it
                D.wait[0] = FusedWait{D.flags2, Tp, Pe, 0, 0, kFusedRing};
                D.wait[2] = FusedWait{fl[rl], Tp, 1, 1, 0, 0};
                D.G = ly.G; D.Wg = ly.WihT; D.xchg_ext = h->fused_xchg_ext[rdx]; D.out = e.dh_ext;
            }
        }
        if ((rc = dev_alloc(h, &h->fused_roles_bwd, (size_t)n_roles))) return rc;
        PL_HIP(hipMemcpyAsync(h->fused_roles_bwd, roles.data(), sizeof(FusedRole) * n_roles, hipMemcpyHostToDevice, h->stream));
        PL_HIP(hipStreamSynchronize(h->stream));
    }
    return PL_OK;
}

// predictor + mel head + embedder LSTM layers as one launch; false = not taken (the caller runs the per-layer path)
bool fused_acoustic_forward(pl_handle* h, hipStream_t st) {
    if (!h->fused_fwd_ok || h->sweep_slot < 0) return false;
    h->pred_dA_skipped = false;   // the forward pass rewrites the gate stash   // the flag slices are zeroed at the top of an iteration only
    launch_pack_cp(st, h->dt, h->x, h->B, h->T, h->C, h->X0, h->Bp, h->Cp);
    FusedArgs a{};
    fused_common_args(h, a, h->fused_grid_fwd, h->fused_tab_fwd, h->fused_roles_fwd, false);
    if (h->fused_rows16) launch_fused_fwd16(st, h->pred.Hp, h->emb.Hp, a);
    else if (h->fused_fwd2) launch_fused_fwd2(st, h->pred.Hp, h->emb.Hp, a);
    else launch_fused_fwd(st, h->pred.Hp, h->emb.Hp, a);
    return true;
}

// embedder layers, their dL/dh products, the backward mel head and the predictor's recurrence as one launch.  Before it: dL/dsem
// -> dL/dh at the embedder's last step (h->dv), and the loss part of dL/dY in f32 (h->Y is free in the fused forward: the mel
// head there pools without storing Y).  After it: dL/dCP = dA W_ih of the predictor, one product.
bool fused_acoustic_backward(pl_handle* h, hipStream_t st, const LossArgs& la) {
    Model& p = h->pred;
    if (!h->fused_bwd_ok || h->sweep_slot < 0) return false;
    launch_dsem(st, h->dt, la, h->dsem);
    emb_head_backward(h, st);
    launch_dy(st, F32, la, nullptr, h->Y);
    FusedArgs a{};
    fused_common_args(h, a, h->fused_grid_bwd, h->fused_tab_bwd, h->fused_roles_bwd, true);
    if (h->fused_rows16) launch_fused_bwd16(st, p.Hp, h->emb.Hp, a);
    else launch_fused_bwd(st, p.Hp, h->emb.Hp, a);
    LstmLayer& l0 = p.layers[0];
    launch_gemm_nt(st, h->dt, true, l0.G, 4 * p.Hp, l0.WihT, 4 * p.Hp, nullptr, h->dX, l0.in_p, h->T * h->Bp, l0.in_p, 4 * p.Hp);
    return true;
}

LossArgs loss_args(pl_handle* h, bool with_sem) {
    LossArgs a{};
    a.B = h->B; a.T = h->T; a.Tp = h->Tp; a.C = h->C; a.M = h->M; a.S = h->S;
    a.Bp = h->Bp; a.Mp = h->Mp; a.Sp = h->Sp;
    a.w_mel = h->cfg.w_mel; a.w_sem = h->cfg.w_sem; a.w_vel = h->cfg.w_vel; a.w_jerk = h->cfg.w_jerk; a.w_ll = h->cfg.w_ll;
    a.use_mel = h->use_mel() ? 1 : 0;
    a.use_sem = with_sem ? 1 : 0;
    a.x = h->x; a.mel = h->mel_bm; a.target_mel = h->target_mel;
    a.sem = with_sem ? h->sem : nullptr;
    a.mel2 = h->tube_on() ? h->mel2_bm : nullptr;
    a.sem2 = h->tube_on() ? h->sem2 : nullptr;
    a.target_sem = h->target_sem;
    a.scal = h->scal; a.loss_rows = h->loss_rows; a.iter_slot = h->counters + 1; a.dwork = h->dwork; a.part = h->loss_part;
    a.cls_wb = h->cls_on ? h->cls_wb : nullptr; a.w_cls = h->w_cls;
    return a;
}

AdamArgs adam_args(pl_handle* h) {
    AdamArgs a{};
    a.B = h->B; a.T = h->T; a.C = h->C; a.Bp = h->Bp; a.Cp = h->Cp;
    a.lr = h->cfg.lr; a.beta1 = h->cfg.beta1; a.beta2 = h->cfg.beta2; a.eps = h->cfg.eps;
    a.clamp_lo = h->cfg.clamp_lo; a.clamp_hi = h->cfg.clamp_hi;
    a.w_vel = h->cfg.w_vel; a.w_jerk = h->cfg.w_jerk; a.w_ll = h->cfg.w_ll;
    a.smiling = h->cfg.smiling;
    a.dX = h->dX; a.dX2 = h->tube_on() ? h->dX2 : nullptr; a.x = h->x; a.m = h->m; a.v = h->v; a.grad = h->grad; a.dwork = h->dwork;
    a.step_count = h->counters; a.iter_slot = h->counters + 1;
    a.past = h->past_len > 0 ? h->past : nullptr;
    a.past_len = h->past_len; a.past_per_utt = h->past_per_utt;
    return a;
}

// one inner iteration: forward, criterion, backward-data, Adam + projection
void enqueue_iteration(pl_handle* h, hipStream_t st) {
    gemm_set_big(h->gemm_big);
    const bool with_sem = h->need_emb_in_step();
    h->wf_next = 0;
    h->wf_stream_next = 0;
    h->pred_dA_skipped = false;
    zero_all_sweep_slots(h, st);   // the flags of all sweeps of the iteration in one launch
    const int pipe_nc = h->fused_fwd_ok ? 0 : acoustic_pipeline_chunks(h);   // the fused launches (batches of 49+ rows) come first
    if (pipe_nc) {
        acoustic_forward_pipeline(h, st, pipe_nc);
    } else if (fused_acoustic_forward(h, st)) {
        emb_head_forward(h, st, nullptr);
    } else {
        pred_forward(h, st);
        if (with_sem) emb_forward(h, st, nullptr, h->mel_bm);
    }
    if (h->stop_after_fwd) {   // diagnostic: leave the forward stashes as they are (tools/fused_check.py)
        h->sweep_slot = -1;
        return;
    }
    const int tube_nc = tube_pipeline_chunks(h);
    if (tube_nc) tube_forward_pipeline(h, st, tube_nc);
    else if (h->tube_on()) tube_forward(h, st);
    LossArgs la = loss_args(h, with_sem);
    launch_loss_reduce(st, la);
    launch_loss_finalize(st, la);
    // the fused backward launch does not care how the stashes were made: it also follows a chunk-pipelined forward pass
    if (with_sem && fused_acoustic_backward(h, st, la)) {
    } else if (pipe_nc) {
        acoustic_backward_pipeline(h, st, pipe_nc, la);
    } else {
        const float* dmel_e = nullptr;
        if (with_sem) {
            launch_dsem(st, h->dt, la, h->dsem);
            emb_backward(h, st);
            dmel_e = h->dmel_e;
        }
        launch_dy(st, h->dt, la, dmel_e, h->dY);
        Model& p = h->pred;
        // dL/dh_top(t) = dY_t * W_p
        launch_gemm_nt(st, h->dt, false, h->dY, h->Mp, p.WlinT, h->Mp, nullptr, p.dh_ext, p.Hp, h->T * h->Bp, p.Hp, h->Mp);
        model_backward(h, st, p, nullptr, h->dX);
    }
    if (tube_nc) tube_backward_pipeline(h, st, tube_nc, la);
    else if (h->tube_on()) tube_backward(h, st, la);
    AdamArgs aa = adam_args(h);
    launch_cp_update(st, aa);
    h->sweep_slot = -1;
}

// PAULE_HIP_DEBUG=segv_trace (diagnostic): print the native frames of a segmentation fault before dying -- the crash inside
// hipGraphLaunch after branched graph execs were destroyed (drop_graph) leaves no other trace
void segv_trace(int sig) {
    void* frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "[pl] SIGSEGV, native frames:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PL_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return PL_OK;
}

static const bool g_dbg_graph = debug_word("graph");
// graph execs with parallel branches that retired handles left allocated (see drop_graph): counted, so that a host can see the leak
// the containment of the runtime crash costs (pl_plan_info: PL_PLAN_RETAINED_EXECS, process-wide)
static std::atomic<int> g_retained_branched_execs{0};
#define DBG_G(msg) do { if (g_dbg_graph) { fprintf(stderr, "[pl] %s\n", msg); fflush(stderr); } } while (0)
void drop_graph(pl_handle* h) {
    if (!h->graph_exec && !h->graph) return;
    // pl_step is asynchronous: a launch of this exec may still be in flight.  Nothing of it may be retired under a running
    // launch -- and the branches of a wavefront graph run on streams of the runtime's own, so for those the whole device is
    // drained, not just the handle's stream (PAULE_HIP_DESTROY_BRANCHED=2: destroy them after that; see below)
    DeviceGuard guard(h->cfg.device);
    (void)hipStreamSynchronize(h->stream);
    if (!h->wf_regions.empty()) (void)hipDeviceSynchronize();
    DBG_G("drop: exec destroy");
    // A graph exec with parallel branches (layer wavefront) is NOT destroyed: after hipGraphExecDestroy of such execs, the first
    // hipGraphLaunch of a later branched exec crashed inside the runtime about once in a hundred handles (ROCm 7.2;
    // tools/microbench/capture_stress.py reproduces it with PAULE_HIP_DESTROY_BRANCHED=1, 720 handles pass without).  The exec
    // of a retired handle stays allocated for the life of the process (kernel arguments only; device buffers are freed).
#ifdef PL_EXPERIMENTS   // PAULE_HIP_DESTROY_BRANCHED reproduces the runtime crash of DESIGN.md section 10 (tools/microbench/capture_stress.py)
    const bool destroy_branched = std::getenv("PAULE_HIP_DESTROY_BRANCHED") != nullptr;
#else
    const bool destroy_branched = false;
#endif
    if (h->graph_exec && (h->wf_regions.empty() || destroy_branched)) (void)hipGraphExecDestroy(h->graph_exec);
    else if (h->graph_exec) g_retained_branched_execs.fetch_add(1, std::memory_order_relaxed);
    DBG_G("drop: graph destroy");
    if (h->graph) (void)hipGraphDestroy(h->graph);
    DBG_G("drop: done");
    h->graph_exec = nullptr;
    h->graph = nullptr;
}

int build_graph(pl_handle* h) {
    if (!h->cap_stream) {   // one capture stream per device for the life of the process, like the layer streams (wf_stream_pool)
        static hipStream_t cap[SweepChain::kMaxDev] = {};
        const int dev = h->cfg.device;
        if (dev < 0 || dev >= SweepChain::kMaxDev) return fail(PL_ERR_INVALID, "device ordinal out of range for graph capture");
        if (!cap[dev]) PL_HIP(hipStreamCreateWithFlags(&cap[dev], hipStreamNonBlocking));
        h->cap_stream = cap[dev];
    }
    DBG_G("capture begin");
    PL_HIP(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeRelaxed));
    h->wf_regions.clear();
    h->capturing = true;
    enqueue_iteration(h, h->cap_stream);
    h->capturing = false;
    DBG_G("end capture");
    hipError_t e = hipStreamEndCapture(h->cap_stream, &h->graph);
    DBG_G("end capture done");
    if (e != hipSuccess) return fail(PL_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    if (!h->wf_regions.empty()) {
        // the capture is one chain; give the wavefront regions their real dependencies: segment (layer, chunk) waits for
        // (layer below / above, same chunk) and (same layer, previous chunk) only, the node after a region for all its final segments
        size_t n_edges = 0;
        PL_HIP(hipGraphGetEdges(h->graph, nullptr, nullptr, &n_edges));
        std::vector<hipGraphNode_t> from(n_edges), to(n_edges);
        if (n_edges) PL_HIP(hipGraphGetEdges(h->graph, from.data(), to.data(), &n_edges));
        std::unordered_map<hipGraphNode_t, hipGraphNode_t> succ;
        for (size_t i = 0; i < n_edges; ++i) succ[from[i]] = to[i];
        for (const pl_handle::WfRegion& rg : h->wf_regions) {
            const size_t ns = rg.segs.size();
            std::vector<hipGraphNode_t> first(ns, nullptr);
            for (size_t k = 0; k < ns; ++k) {
                auto it = succ.find(rg.segs[k].before);
                if (rg.segs[k].before == nullptr || it == succ.end() || rg.segs[k].last == nullptr)
                    return fail(PL_ERR_HIP, "graph capture of the layer wavefront: unexpected node chain");
                first[k] = it->second;
            }
            auto after = succ.find(rg.segs[ns - 1].last);
            for (size_t k = 1; k < ns; ++k) {
                const pl_handle::WfSeg& sg = rg.segs[k];
                hipGraphNode_t deps[2];
                size_t nd = 0;
                bool keep_chain = false;
                for (int d : {sg.dep1, sg.dep2})
                    if (d >= 0) {
                        if (rg.segs[d].last == sg.before) keep_chain = true;
                        else deps[nd++] = rg.segs[d].last;
                    }
                if (!keep_chain) PL_HIP(hipGraphRemoveDependencies(h->graph, &sg.before, &first[k], 1));
                for (size_t i = 0; i < nd; ++i) PL_HIP(hipGraphAddDependencies(h->graph, &deps[i], &first[k], 1));
            }
            if (after != succ.end())
                for (int f : rg.finals)
                    if ((size_t)f != ns - 1) PL_HIP(hipGraphAddDependencies(h->graph, &rg.segs[f].last, &after->second, 1));
        }
    }
    DBG_G("instantiate");
    PL_HIP(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
    DBG_G("instantiate done");
    return PL_OK;
}

double lstm_flops_per_step(int L, int H, int in) {
    double f = 0;
    for (int l = 0; l < L; ++l) f += 2.0 * 4 * H * ((l == 0 ? in : H) + H);
    return f;
}

}  // namespace

// =================================================================================================
// C-ABI
// =================================================================================================
extern "C" {

const char* pl_last_error(void) { return g_last_error.c_str(); }
int pl_version(void) { return PL_VERSION; }
int pl_hip_version_built(void) { return HIP_VERSION; }

int pl_default_config(pl_config* cfg) {
    if (!cfg) return fail(PL_ERR_INVALID, "cfg is NULL");
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int32_t)sizeof(pl_config);
    cfg->batch = 1;
    cfg->n_frames = 0;
    cfg->cp_dim = 30;
    cfg->mel_dim = 60;
    cfg->sem_dim = 300;
    cfg->pred_layers = 1;     // Paule default predictive model (paule/paule.py:124)
    cfg->pred_hidden = 720;
    cfg->emb_layers = 2;      // Paule default embedder (paule/paule.py:167)
    cfg->emb_hidden = 720;
    cfg->dtype = PL_F32;
    cfg->objective = PL_OBJ_ACOUSTIC;
    cfg->w_mel = 5.0f;        // paule/paule.py:592-597
    cfg->w_sem = 10.0f;
    cfg->w_vel = 80.0f;
    cfg->w_jerk = 400.0f;
    cfg->w_ll = 100000.0f;
    cfg->lr = 0.01f;
    cfg->beta1 = 0.9f;
    cfg->beta2 = 0.999f;
    cfg->eps = 1e-8f;
    cfg->clamp_lo = -1.05f;
    cfg->clamp_hi = 1.05f;
    cfg->smiling = 0;
    cfg->device = 0;
    cfg->use_graph = 1;
    cfg->stream = nullptr;
    cfg->inv_layers = 0;      // no inverse model unless asked for (Paule's: 1 x 720, paule/paule.py:146)
    cfg->inv_hidden = 720;
    cfg->inv_mel_blocks = 3;  // paule/models.py:190, :194
    cfg->inv_res_blocks = 5;
    return PL_OK;
}

int pl_create(const pl_config* cfg, pl_handle** out) {
    if (!cfg || !out) return fail(PL_ERR_INVALID, "pl_create: NULL argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(pl_config))
        return fail(PL_ERR_INVALID, "pl_create: pl_config.struct_size mismatch (ABI drift)");
    if (cfg->batch < 1 || cfg->cp_dim < 1 || cfg->mel_dim < 1 || cfg->sem_dim < 1)
        return fail(PL_ERR_INVALID, "pl_create: batch and feature dims must be >= 1");
    if (cfg->n_frames < 14)
        return fail(PL_ERR_INVALID, "pl_create: n_frames must be >= 14 (jerk needs T - 12 >= 1 frames, mel needs T/2 >= 1)");
    if (cfg->pred_layers < 1 || cfg->pred_hidden < 1) return fail(PL_ERR_INVALID, "pl_create: predictive model needs >= 1 LSTM layer");
    if (cfg->emb_layers < 0 || (cfg->emb_layers > 0 && cfg->emb_hidden < 1)) return fail(PL_ERR_INVALID, "pl_create: bad embedder shape");
    if (cfg->emb_post_size < 0 || cfg->emb_mel_blocks < 0 || cfg->emb_mel_blocks > 16 ||
        (cfg->emb_layers == 0 && (cfg->emb_post_size > 0 || cfg->emb_mel_blocks > 0)))
        return fail(PL_ERR_INVALID, "pl_create: bad embedder head / mel block shape");
    if (cfg->cp_tube_layers < 0 || cfg->tube_mel_layers < 0 || cfg->tube_emb_layers < 0) return fail(PL_ERR_INVALID, "pl_create: bad tube model shape");
    if (cfg->cp_tube_layers > 0) {
        if (cfg->tube_dim < 1 || cfg->cp_tube_hidden < 1 || cfg->tube_mel_layers < 1 || cfg->tube_mel_hidden < 1 || cfg->tube_emb_layers < 1 ||
            cfg->tube_emb_hidden < 1)
            return fail(PL_ERR_INVALID, "pl_create: somatosensory feedback needs all three tube models (cp_tube, tube_mel, tube_emb)");
        if (cfg->emb_layers == 0 || cfg->objective == PL_OBJ_ACOUSTIC)
            return fail(PL_ERR_INVALID, "pl_create: somatosensory feedback runs with the objectives acoustic_semvec and semvec only (the "
                                        "reference's acoustic criterion fails there, paule/paule.py:692)");
    }
    if (cfg->emb_mel_blocks > 0 && cfg->mel_dim % 3 != 0) return fail(PL_ERR_INVALID, "pl_create: mel blocks need mel_dim divisible by 3");
    if (cfg->dtype != PL_F32 && cfg->dtype != PL_BF16) return fail(PL_ERR_INVALID, "pl_create: dtype must be PL_F32 or PL_BF16");
    if (cfg->objective < PL_OBJ_ACOUSTIC || cfg->objective > PL_OBJ_SEMVEC)
        return fail(PL_ERR_INVALID, "objective has to be one of 'acoustic_semvec', 'acoustic' or 'semvec'");
    if (cfg->objective != PL_OBJ_ACOUSTIC && cfg->emb_layers == 0)
        return fail(PL_ERR_INVALID, "pl_create: semvec objectives need an embedder (emb_layers > 0)");
    if (cfg->cp_dim < 5 && cfg->smiling) return fail(PL_ERR_INVALID, "pl_create: smiling needs cp_dim >= 5");
    if (cfg->inv_layers < 0 || (cfg->inv_layers > 0 && (cfg->inv_hidden < 1 || cfg->inv_mel_blocks < 0 || cfg->inv_res_blocks < 0 ||
                                                        cfg->inv_mel_blocks > 64 || cfg->inv_res_blocks > 64)))
        return fail(PL_ERR_INVALID, "pl_create: bad inverse-model shape");
    if (cfg->inv_layers > 0 && cfg->inv_mel_blocks > 0 && cfg->mel_dim % 3 != 0)
        return fail(PL_ERR_INVALID, "pl_create: MelChannelConv1D needs mel_dim divisible by 3 (paule/models.py:146)");
    int ndev = 0;
    PL_HIP(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(PL_ERR_INVALID, "pl_create: no such HIP device");
    DeviceGuard guard(cfg->device);

    if (debug_word("segv_trace")) signal(SIGSEGV, segv_trace);
    pl_handle* h = new pl_handle();
    h->cfg = *cfg;
    h->stream = static_cast<hipStream_t>(cfg->stream);
    h->B = cfg->batch; h->T = cfg->n_frames; h->Tp = cfg->n_frames / 2;
    h->C = cfg->cp_dim; h->M = cfg->mel_dim; h->S = cfg->sem_dim;
    h->Bp = pad16(h->B); h->Cp = pad32(h->C); h->Mp = pad32(h->M); h->Sp = pad32(h->S);
    h->dt = cfg->dtype == PL_BF16 ? BF16 : F32;
    h->act = dtype_size(h->dt);
    h->loss_cap = 256;
    h->rows_in_use = h->B;

    int rc = PL_OK;
    auto bail = [&](int code) { pl_destroy(h); return code; };
    if ((rc = alloc_model(h, h->pred, cfg->pred_layers, cfg->pred_hidden, h->C, h->M, h->T))) return bail(rc);
    int hmax = h->pred.Hp;
    if (cfg->emb_layers > 0) {
        h->emb_post = cfg->emb_post_size;
        h->emb_post_p = pad32(cfg->emb_post_size);
        h->emb_blocks = cfg->emb_mel_blocks;
        if ((rc = alloc_model(h, h->emb, cfg->emb_layers, cfg->emb_hidden, h->M, h->emb_post > 0 ? h->emb_post : h->S, h->Tp))) return bail(rc);
        hmax = hmax > h->emb.Hp ? hmax : h->emb.Hp;
    }
    const size_t Bp = h->Bp, T = h->T, Tp = h->Tp, B = h->B;
    if (cfg->cp_tube_layers > 0) {
        h->U = cfg->tube_dim;
        h->Up = pad32(cfg->tube_dim);
        if ((rc = alloc_model(h, h->tube, cfg->cp_tube_layers, cfg->cp_tube_hidden, h->C, h->U, h->T))) return bail(rc);
        if ((rc = alloc_model(h, h->tmel, cfg->tube_mel_layers, cfg->tube_mel_hidden, h->U, h->M, h->T))) return bail(rc);
        if ((rc = alloc_model(h, h->temb, cfg->tube_emb_layers, cfg->tube_emb_hidden, h->U, h->S, h->T))) return bail(rc);
        for (Model* md : {&h->tube, &h->tmel, &h->temb}) hmax = hmax > md->Hp ? hmax : md->Hp;
        const size_t Up = h->Up;
        if ((rc = alloc_act(h, &h->tube_tm, T * Bp * Up))) return bail(rc);
        if ((rc = dev_alloc(h, &h->Y2, T * Bp * h->Mp))) return bail(rc);
        if ((rc = dev_alloc(h, &h->mel2_bm, Bp * Tp * h->M))) return bail(rc);
        if ((rc = alloc_act(h, &h->mel2_tm, Tp * Bp * h->Mp))) return bail(rc);
        if ((rc = alloc_act(h, &h->h_last2, Bp * h->temb.Hp))) return bail(rc);
        if ((rc = dev_alloc(h, &h->sem2, Bp * h->Sp))) return bail(rc);
        if ((rc = alloc_act(h, &h->dsem2, Bp * h->Sp))) return bail(rc);
        if ((rc = alloc_act(h, &h->dv2, Bp * h->temb.Hp))) return bail(rc);
        if ((rc = alloc_act(h, &h->dY2, T * Bp * h->Mp))) return bail(rc);
        if ((rc = alloc_act(h, &h->dYt, T * Bp * Up))) return bail(rc);
        if ((rc = dev_alloc(h, &h->dtube_a, T * Bp * Up))) return bail(rc);
        if ((rc = dev_alloc(h, &h->dtube_b, T * Bp * Up))) return bail(rc);
        if ((rc = dev_alloc(h, &h->dX2, T * Bp * h->Cp))) return bail(rc);
    }
    if (cfg->inv_layers > 0) {
        if ((rc = alloc_model(h, h->inv, cfg->inv_layers, cfg->inv_hidden, 3 * h->M, h->C, h->Tp))) return bail(rc);
        hmax = hmax > h->inv.Hp ? hmax : h->inv.Hp;
        h->inv_mel_blocks = cfg->inv_mel_blocks;
        h->inv_res_blocks = cfg->inv_res_blocks;
        const size_t G = h->M / 3, C = h->C;
        const size_t n_conv = (size_t)h->inv_mel_blocks * 3 * G * 16 + (size_t)h->inv_res_blocks * 2 * C * 6 + C * 11;
        if ((rc = dev_alloc(h, &h->inv_conv, n_conv))) return bail(rc);
        h->inv_conv_set.assign((size_t)h->inv_mel_blocks * 3 + (size_t)h->inv_res_blocks * 2 + 1, 0);
        for (int i = 0; i < 2; ++i)
            if ((rc = dev_alloc(h, &h->inv_x[i], B * Tp * h->M))) return bail(rc);
        if ((rc = alloc_act(h, &h->inv_in, Tp * Bp * h->inv.in_p))) return bail(rc);
        if ((rc = dev_alloc(h, &h->inv_Y, Tp * Bp * h->Cp))) return bail(rc);
        for (int i = 0; i < 3; ++i)
            if ((rc = dev_alloc(h, &h->inv_z[i], B * 2 * Tp * h->C))) return bail(rc);
    }
    for (int i = 0; i < 2; ++i) {
        if ((rc = dev_alloc(h, &h->c_run[i], Bp * hmax))) return bail(rc);
        if ((rc = dev_alloc(h, &h->dc_run[i], Bp * hmax))) return bail(rc);
    }
    if ((rc = alloc_act(h, &h->X0, T * Bp * h->Cp))) return bail(rc);
    if ((rc = dev_alloc(h, &h->Y, T * Bp * h->Mp))) return bail(rc);
    if ((rc = dev_alloc(h, &h->mel_bm, Bp * Tp * h->M))) return bail(rc);   // Bp rows: a training mini-batch may use all padded rows
    if ((rc = alloc_act(h, &h->mel_tm, Tp * Bp * h->Mp))) return bail(rc);
    if ((rc = alloc_act(h, &h->dY, T * Bp * h->Mp))) return bail(rc);
    if ((rc = dev_alloc(h, &h->dX, T * Bp * h->Cp))) return bail(rc);
    if (cfg->emb_layers > 0) {
        if (h->emb_post > 0) {
            const size_t Pp = h->emb_post_p;
            if ((rc = alloc_act(h, &h->Wup, (size_t)h->Sp * Pp))) return bail(rc);
            if ((rc = alloc_act(h, &h->WupT, Pp * h->Sp))) return bail(rc);
            if ((rc = dev_alloc(h, &h->bup, h->Sp))) return bail(rc);
            if ((rc = dev_alloc(h, &h->head_pre, Bp * Pp))) return bail(rc);
            if ((rc = alloc_act(h, &h->head_act, Bp * Pp))) return bail(rc);
            if ((rc = dev_alloc(h, &h->head_d, Bp * Pp))) return bail(rc);
        }
        if (h->emb_blocks > 0) {
            const size_t G = h->M / 3;
            if ((rc = dev_alloc(h, &h->emb_conv, (size_t)h->emb_blocks * 3 * G * 16))) return bail(rc);
            h->emb_conv_set.assign((size_t)h->emb_blocks * 3, 0);
            for (int i = 0; i < 2; ++i)
                if ((rc = dev_alloc(h, &h->emb_x[i], B * Tp * h->M))) return bail(rc);
            if ((rc = alloc_act(h, &h->emb_in, Tp * Bp * h->Mp))) return bail(rc);
            if ((rc = dev_alloc(h, &h->emb_din, Tp * Bp * h->Mp))) return bail(rc);
        }
        if ((rc = alloc_act(h, &h->h_last, Bp * h->emb.Hp))) return bail(rc);
        if ((rc = dev_alloc(h, &h->sem, Bp * h->Sp))) return bail(rc);
        if ((rc = alloc_act(h, &h->dsem, Bp * h->Sp))) return bail(rc);
        if ((rc = alloc_act(h, &h->dv, Bp * h->emb.Hp))) return bail(rc);
        if ((rc = dev_alloc(h, &h->dmel_e, Tp * Bp * h->Mp))) return bail(rc);
        if ((rc = dev_alloc(h, &h->target_sem, B * h->S))) return bail(rc);
        if ((rc = dev_alloc(h, &h->out_tmp, B * h->S))) return bail(rc);
    }
    if ((rc = dev_alloc(h, &h->x, B * T * h->C))) return bail(rc);
    if ((rc = dev_alloc(h, &h->m, B * T * h->C))) return bail(rc);
    if ((rc = dev_alloc(h, &h->v, B * T * h->C))) return bail(rc);
    if ((rc = dev_alloc(h, &h->grad, B * T * h->C))) return bail(rc);
    if ((rc = dev_alloc(h, &h->dwork, 3 * B * T * h->C))) return bail(rc);
    if ((rc = dev_alloc(h, &h->target_mel, B * Tp * h->M))) return bail(rc);
    if ((rc = dev_alloc(h, &h->cls_wb, h->M + 1))) return bail(rc);
    if ((rc = dev_alloc(h, &h->scal, B * 8))) return bail(rc);
    if ((rc = dev_alloc(h, &h->loss_part, B * (size_t)loss_chunks(T) * 8))) return bail(rc);
    if ((rc = dev_alloc(h, &h->loss_rows, (size_t)h->loss_cap * B * PL_LOSS_COLS))) return bail(rc);
    if ((rc = dev_alloc(h, &h->counters, 4))) return bail(rc);
    {
        hipDeviceProp_t prop;
        hipError_t pe = hipGetDeviceProperties(&prop, cfg->device);
        if (pe != hipSuccess) { bail(PL_ERR_HIP); return fail(PL_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(pe)); }
        h->n_cu = prop.multiProcessorCount;
        const char* env = std::getenv("PAULE_HIP_NO_SWEEP");
        h->use_sweep = !(env && env[0] == '1');
#ifdef PL_EXPERIMENTS   // A/B-only switch (its measured winner is the fixed default of the shipped library: DESIGN.md 4.1c)
        if (const char* z = std::getenv("PAULE_HIP_ZERO_MODE")) h->zero_mode = std::atoi(z);
#endif
        if (const char* z = std::getenv("PAULE_HIP_BWD_MODE")) h->bwd_mode = std::atoi(z);
        if (const char* z = std::getenv("PAULE_HIP_BWD_WAVES")) h->bwd_waves = std::atoi(z) == 4 ? 4 : 8;
        if (const char* z = std::getenv("PAULE_HIP_BWD_STREAM")) h->bwd_stream = std::atoi(z) != 0 ? 1 : 0;
        if (const char* z = std::getenv("PAULE_HIP_FUSED_XCD")) h->fused_xcd = std::atoi(z) != 0;
        if (const char* z = std::getenv("PAULE_HIP_FUSED2_XCD")) h->fused2_xcd = std::atoi(z) != 0 && fused_fwd2_xcd_compiled();
        if (const char* z = std::getenv("PAULE_HIP_FUSED_OCC2")) h->fused_occ2 = std::atoi(z);
        if (const char* z = std::getenv("PAULE_HIP_SWEEP2")) h->sweep2 = std::atoi(z);
        if (const char* z = std::getenv("PAULE_HIP_BWD_XT")) h->bwd_xt = std::atoi(z);
        if (const char* z = std::getenv("PAULE_HIP_BWD_CHAINS")) h->bwd_chains = std::atoi(z);
        if (const char* z = std::getenv("PAULE_HIP_BWD_PF")) h->bwd_pf = std::atoi(z);
        if (const char* z = std::getenv("PAULE_HIP_FUSED2_PRIO")) h->fused2_prio = std::atoi(z);
        if (const char* z = std::getenv("PAULE_HIP_FUSED16_PF")) h->fused16_pf = std::atoi(z);
        if (const char* z = std::getenv("PAULE_HIP_F32_STREAM")) h->f32_stream = std::atoi(z);
        if (const char* z = std::getenv("PAULE_HIP_BWD_PF_DIST")) h->bwd_pf_dist = std::atoi(z);
#ifdef PL_EXPERIMENTS   // round 4's hand-off experiments (profiles/r04_token_handoff.txt): not in the shipped library
        if (const char* z = std::getenv("PAULE_HIP_BWD_DMA")) h->bwd_dma = std::atoi(z) & 3;
        if (const char* z = std::getenv("PAULE_HIP_TOKEN_EARLY")) h->token_early = std::atoi(z) != 0;
        if (const char* z = std::getenv("PAULE_HIP_BWD_STREAM")) h->bwd_stream = std::atoi(z) < 0 ? 0 : (std::atoi(z) > 2 ? 2 : std::atoi(z));
#endif
        if (const char* z = std::getenv("PAULE_HIP_TN_BF16")) h->tn_bf16 = std::atoi(z) != 0;
        h->debug_fused = debug_word("fused");
        if (const char* z = std::getenv("PAULE_HIP_GEMM_BIG")) h->gemm_big = std::atoi(z) != 0;
        gemm_big_init();
        h->census_hooks = debug_word("census_expect_extra") || debug_word("census_late_ms");
        if (debug_word("stop_after_fwd")) {
            h->stop_after_fwd = true;
            fprintf(stderr, "[pl] PAULE_HIP_DEBUG=stop_after_fwd: pl_step of this handle enqueues the forward pass ONLY (diagnostic mode: no loss, "
                            "no backward pass, no update -- loss logs and the CP do not change)\n");
        }
        if (const char* z = std::getenv("PAULE_HIP_XCD_FAST")) h->xcd_fast = std::atoi(z);
#ifdef PL_EXPERIMENTS   // A/B-only switch (its measured winner is the fixed default of the shipped library: DESIGN.md 4.1c)
        if (const char* z = std::getenv("PAULE_HIP_POLL_MASK")) h->poll_mask = (unsigned)std::atoi(z);
        if (const char* z = std::getenv("PAULE_HIP_FUSE_INPUT")) h->fuse_input = std::atoi(z) != 0;
#endif
        if (const char* z = std::getenv("PAULE_HIP_F32_SWEEP")) h->f32_sweep = std::atoi(z) != 0;
        if (const char* z = std::getenv("PAULE_HIP_F32_CHAINS")) h->f32_chains = std::atoi(z);
#ifdef PL_EXPERIMENTS   // A/B-only switch (its measured winner is the fixed default of the shipped library: DESIGN.md 4.1c)
        if (const char* z = std::getenv("PAULE_HIP_SMALL_GRID")) h->small_grid = std::atoi(z) != 0;
#endif
        if (const char* z = std::getenv("PAULE_HIP_SWEEP16")) h->sweep16 = std::atoi(z) != 0;
#ifdef PL_EXPERIMENTS   // A/B-only switch (its measured winner is the fixed default of the shipped library: DESIGN.md 4.1c)
        if (const char* z = std::getenv("PAULE_HIP_WIDE_INGEST16")) h->wide_ingest16 = std::atoi(z) != 0;
        if (const char* z = std::getenv("PAULE_HIP_WIDE_INGEST")) h->wide_ingest = std::atoi(z) != 0;
        if (const char* z = std::getenv("PAULE_HIP_OWN_STORE")) h->own_store = std::atoi(z) != 0;
        if (const char* z = std::getenv("PAULE_HIP_XCD_FAST16")) h->xcd_fast16 = std::atoi(z) != 0;
        if (const char* z = std::getenv("PAULE_HIP_STASH_LDS")) h->stash_lds = std::atoi(z) != 0;
#endif
        if (h->dt == F32 && h->use_sweep && h->f32_sweep) {
            size_t xb = 0;
            for (Model* md : {&h->pred, &h->emb, &h->tube, &h->tmel, &h->temb})
                if (md->L > 0 && lstm_sweep_f32_supported(md->Hp)) {
                    const size_t xe = lstm_f32_exchange_bytes(md->Hp, h->Bp);
                    xb = xb > xe ? xb : xe;
                }
            if (xb && (rc = raw_alloc(h, &h->sweep_xchg, xb))) return bail(rc);
        }
        if (h->dt == BF16 && h->use_sweep && h->bwd_mode == 1) {
            size_t xb = 0;
            for (Model* md : {&h->pred, &h->emb, &h->tube, &h->tmel, &h->temb})
                if (md->L > 0 && lstm_sweep_supported(h->dt, md->Hp)) {
                    const size_t xe = lstm_rs_exchange_bytes(md->Hp, h->Bp);
                    xb = xb > xe ? xb : xe;
                }
            if (xb && (rc = raw_alloc(h, &h->sweep_xchg, xb))) return bail(rc);
            if (xb && h->bwd_stream == 2) {   // the token form's own exchange (raw_alloc zeroes it: every granule "retired")
                if ((rc = raw_alloc(h, &h->sweep_xchg_tok, xb))) return bail(rc);
                h->sweep_xchg_tok_bytes = xb;
            }
        }
        if (const char* z = std::getenv("PAULE_HIP_F32_VALU")) h->f32_valu = std::atoi(z) != 0;
        if (const char* z = std::getenv("PAULE_HIP_WF_PIPELINE")) h->wf_pipeline = std::atoi(z) != 0;
#ifdef PL_EXPERIMENTS   // A/B-only switch (its measured winner is the fixed default of the shipped library: DESIGN.md 4.1c)
        if (const char* z = std::getenv("PAULE_HIP_PIPE_SPREAD")) h->pipe_spread = std::atoi(z) != 0;
#endif
        if (const char* z = std::getenv("PAULE_HIP_WAVEFRONT")) h->wavefront = std::atoi(z);
#ifdef PL_EXPERIMENTS   // A/B-only switch (its measured winner is the fixed default of the shipped library: DESIGN.md 4.1c)
        if (const char* z = std::getenv("PAULE_HIP_WF_DIRS")) h->wf_dirs = std::atoi(z);
#endif
        if (h->wavefront > 0 && h->use_sweep) {
            for (Model* md : {&h->pred, &h->emb, &h->tube, &h->tmel, &h->temb}) {
                const int ppx = pipe_per_xcd(h, *md);
                const bool pipe = cfg->emb_layers > 0 && ppx > 0 && 2 * ppx <= h->n_cu / 8;
                if ((md->L < 2 || wavefront_depth(h, *md) < 2) && !pipe) continue;
                const size_t xb = h->dt == F32 ? (lstm_sweep_f32_supported(md->Hp) ? lstm_f32_exchange_bytes(md->Hp, h->Bp) : 0)
                                               : (lstm_sweep_supported(h->dt, md->Hp) ? lstm_rs_exchange_bytes(md->Hp, h->Bp) : 0);
                if (!xb) continue;
                for (auto& ly : md->layers) {
                    if ((rc = dev_alloc(h, &ly.carry_f, Bp * md->Hp))) return bail(rc);
                    if ((rc = dev_alloc(h, &ly.carry_b, Bp * md->Hp))) return bail(rc);
                    if ((rc = raw_alloc(h, &ly.xchg, xb))) return bail(rc);
                }
            }
            h->wf_on = true;
            // streams and events of an iteration up front (none is created while a capture is running)
            size_t n_streams = 0, n_events = 0;
            for (Model* md : {&h->pred, &h->emb, &h->tube, &h->tmel, &h->temb})
                if (md->L >= 1 && md->layers[0].carry_f) {
                    n_streams += 2 * (size_t)md->L;
                    n_events += 2 * (size_t)md->L * (size_t)(h->wavefront < 32 ? h->wavefront : 32);
                }
            if (cfg->device >= 0 && cfg->device < SweepChain::kMaxDev) {
                std::lock_guard<std::mutex> lock(SweepChain::mu(cfg->device));
                for (size_t i = 0; i < n_streams; ++i) (void)wavefront_stream(h);
                for (size_t i = 0; i < n_events; ++i) (void)wavefront_event(h);
            } else {
                h->wf_on = false;
            }
            h->wf_stream_next = 0;
            h->wf_next = 0;
        }
        if (const char* ms = std::getenv("PAULE_HIP_SPIN_MS")) h->spin_ticks = 100000ull * (unsigned long long)std::atoll(ms);
        if (long long ms = 0; debug_value("census_ms", &ms)) h->census_ticks = 100000ull * (unsigned long long)ms;
        const size_t n_groups_max = (Bp + 7) / 8;   // groups hold >= 8 rows
        const int slice = h->dt == F32 ? 16 : 32;   // hidden units per workgroup
        int pmax = h->pred.Hp / slice;
        if (cfg->emb_layers > 0 && h->emb.Hp / slice > pmax) pmax = h->emb.Hp / slice;
        if (cfg->inv_layers > 0 && h->inv.Hp / slice > pmax) pmax = h->inv.Hp / slice;
        for (Model* md : {&h->tube, &h->tmel, &h->temb})
            if (md->L > 0 && md->Hp / slice > pmax) pmax = md->Hp / slice;
        h->flag_stride = (pmax + 15) / 16 * 16;
        // arrival flags, then the XCD-id table, then (bf16) the per-tile flags of the streamed backward hand-off [2][groups][P][32]
        // (f32: [2][groups][P][64] for lstm_bwd_stream_f32_kernel)
        const size_t n = n_groups_max * T * h->flag_stride + n_groups_max * 64 + 2 * n_groups_max * (size_t)pmax * (h->dt == BF16 ? 32 : 64);
        h->sweep_cnt_bytes = (n * sizeof(int) + 15) / 16 * 16;
        // forward + backward sweep of every layer of an iteration, + the head / projection roles of the fused launches
        h->n_sweep_slots = 2 * (cfg->pred_layers + cfg->emb_layers + cfg->cp_tube_layers + cfg->tube_mel_layers + cfg->tube_emb_layers) +
                           (cfg->emb_layers > 0 ? 2 * fused_roles_count(cfg->pred_layers, cfg->emb_layers) + (cfg->emb_layers - 1) + (cfg->pred_layers - 1) : 0);
        if ((rc = dev_alloc(h, &h->sweep_cnt, h->sweep_cnt_bytes / sizeof(int) * h->n_sweep_slots))) return bail(rc);
        if ((rc = dev_alloc(h, &h->sweep_status, 4))) return bail(rc);
#ifdef PL_STAMPS
        if ((rc = dev_alloc(h, &h->sweep_stamps, 2 * 256 * 8))) return bail(rc);
#endif
    }
    if ((rc = dev_alloc(h, &h->past, B * T * h->C))) return bail(rc);
    if ((rc = plan_fused(h))) return bail(rc);
    // scratch of the ride-along input gradient (lstm_persist_rs.hip, XT): only where the predictor's backward pass really is the streamed
    // 32-row per-layer sweep -- not under the fused backward launch, not on the 16-row kernels (226 MB at cfg3, 754 MB at
    // 128 x 2000 frames, which runs the fused backward, 1.8 GB at 2048 x 300)
    if (h->dt == BF16 && h->use_sweep && h->bwd_mode == 1 && h->sweep_xchg && h->bwd_stream == 1 && h->bwd_waves != 4 && h->bwd_xt && h->pred.L >= 1 &&
        lstm_rs_ride_along_supported(h->pred.Hp, h->pred.layers[0].in_p) && !h->fused_bwd_ok && !use_sweep16(h, h->pred.Hp, true)) {
        void* q = nullptr;
        if ((rc = raw_alloc(h, &q, lstm_rs_xpart_bytes(h->pred.Hp, h->Bp, h->T)))) return bail(rc);
        h->dx_part = static_cast<float*>(q);
    }
    if ((rc = build_fused_roles(h))) return bail(rc);
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { bail(PL_ERR_HIP); return fail(PL_ERR_HIP, std::string("pl_create: ") + hipGetErrorString(e)); }
    *out = h;
    return PL_OK;
}

int pl_destroy(pl_handle* h) {
    if (!h) return PL_OK;
    DeviceGuard guard(h->cfg.device);
    (void)hipStreamSynchronize(h->stream);
    drop_graph(h);
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
    return PL_OK;
}

int pl_set_lstm_weights(pl_handle* h, int model_id, int layer, const float* w_ih, const float* w_hh, const float* b_ih,
                        const float* b_hh) {
    if (!h || !w_ih || !w_hh || !b_ih || !b_hh) return fail(PL_ERR_INVALID, "pl_set_lstm_weights: NULL argument");
    if (model_id < PL_MODEL_PRED || model_id > PL_MODEL_TUBE_EMBED) return fail(PL_ERR_INVALID, "pl_set_lstm_weights: bad model_id");
    Model& md = *model_by_id(h, model_id);
    if (layer < 0 || layer >= md.L) return fail(PL_ERR_INVALID, "pl_set_lstm_weights: layer out of range for this model");
    DeviceGuard guard(h->cfg.device);
    LstmLayer& ly = md.layers[layer];
    hipStream_t st = h->stream;
    launch_pack_matrix(st, h->dt, w_ih, 4, md.H, ly.in, ly.Wih, md.Hp, ly.in_p, false);
    launch_pack_matrix(st, h->dt, w_ih, 4, md.H, ly.in, ly.WihT, md.Hp, ly.in_p, true);
    launch_pack_matrix(st, h->dt, w_hh, 4, md.H, md.H, ly.Whh, md.Hp, md.Hp, false);
    launch_pack_matrix(st, h->dt, w_hh, 4, md.H, md.H, ly.WhhT, md.Hp, md.Hp, true);
    launch_pack_bias(st, b_ih, b_hh, 4, md.H, ly.bias, md.Hp);
    launch_f32_to_f64(st, w_ih, ly.p_wih.x, (int64_t)ly.p_wih.n);
    launch_f32_to_f64(st, w_hh, ly.p_whh.x, (int64_t)ly.p_whh.n);
    launch_f32_to_f64(st, b_ih, ly.p_bih.x, (int64_t)ly.p_bih.n);
    launch_f32_to_f64(st, b_hh, ly.p_bhh.x, (int64_t)ly.p_bhh.n);
    ly.set = true;
    int rc = check_launch();
    if (rc) return rc;
    PL_HIP(hipStreamSynchronize(st));   // caller may free/mutate its tensors on return
    return PL_OK;
}

int pl_set_linear(pl_handle* h, int model_id, const float* w, const float* b) {
    if (!h || !w || !b) return fail(PL_ERR_INVALID, "pl_set_linear: NULL argument");
    if (model_id < PL_MODEL_PRED || model_id > PL_MODEL_TUBE_EMBED) return fail(PL_ERR_INVALID, "pl_set_linear: bad model_id");
    Model& md = *model_by_id(h, model_id);
    if (md.L == 0) return fail(PL_ERR_INVALID, "pl_set_linear: this handle has no such model");
    DeviceGuard guard(h->cfg.device);
    hipStream_t st = h->stream;
    launch_pack_matrix(st, h->dt, w, 1, md.out, md.H, md.Wlin, md.out_p, md.Hp, false);
    launch_pack_matrix(st, h->dt, w, 1, md.out, md.H, md.WlinT, md.out_p, md.Hp, true);
    launch_pack_bias(st, b, nullptr, 1, md.out, md.blin, md.out_p);
    launch_f32_to_f64(st, w, md.p_wlin.x, (int64_t)md.p_wlin.n);
    launch_f32_to_f64(st, b, md.p_blin.x, (int64_t)md.p_blin.n);
    md.lin_set = true;
    int rc = check_launch();
    if (rc) return rc;
    PL_HIP(hipStreamSynchronize(st));
    return PL_OK;
}

int pl_set_speech_classifier(pl_handle* h, const float* w, const float* b, float weight) {
    if (!h) return fail(PL_ERR_INVALID, "pl_set_speech_classifier: NULL handle");
    const bool on = w != nullptr;
    if (on && !b) return fail(PL_ERR_INVALID, "pl_set_speech_classifier: bias is NULL");
    DeviceGuard guard(h->cfg.device);
    if (on) {
        PL_HIP(hipMemcpyAsync(h->cls_wb, w, sizeof(float) * h->M, hipMemcpyDeviceToDevice, h->stream));
        PL_HIP(hipMemcpyAsync(h->cls_wb + h->M, b, sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        PL_HIP(hipStreamSynchronize(h->stream));
    }
    if (on != h->cls_on || weight != h->w_cls) drop_graph(h);   // kernel arguments change
    h->cls_on = on;
    h->w_cls = weight;
    return PL_OK;
}

int pl_set_targets(pl_handle* h, const float* target_mel, const float* target_semvec) {
    if (!h || !target_mel) return fail(PL_ERR_INVALID, "pl_set_targets: target_mel is NULL");
    DeviceGuard guard(h->cfg.device);
    PL_HIP(hipMemcpyAsync(h->target_mel, target_mel, sizeof(float) * h->B * h->Tp * h->M, hipMemcpyDeviceToDevice, h->stream));
    h->have_targets = true;
    if (target_semvec) {
        if (!h->target_sem) return fail(PL_ERR_INVALID, "pl_set_targets: target_semvec given but the handle has no embedder");
        PL_HIP(hipMemcpyAsync(h->target_sem, target_semvec, sizeof(float) * h->B * h->S, hipMemcpyDeviceToDevice, h->stream));
        h->have_sem_target = true;
    }
    PL_HIP(hipStreamSynchronize(h->stream));
    return PL_OK;
}

int pl_set_cp(pl_handle* h, const float* cp) {
    if (!h || !cp) return fail(PL_ERR_INVALID, "pl_set_cp: NULL argument");
    DeviceGuard guard(h->cfg.device);
    launch_f32_to_f64(h->stream, cp, h->x, (int64_t)h->B * h->T * h->C);
    int rc = check_launch();
    if (rc) return rc;
    PL_HIP(hipStreamSynchronize(h->stream));
    h->have_cp = true;
    return PL_OK;
}

int pl_set_past_cp(pl_handle* h, const float* past_cp, int past_len, int per_utterance) {
    if (!h) return fail(PL_ERR_INVALID, "pl_set_past_cp: NULL handle");
    if (!past_cp || past_len <= 0) {
        if (h->past_len != 0) drop_graph(h);
        h->past_len = 0;
        return PL_OK;
    }
    if (past_len % 2 != 0)   // paule/paule.py:575-576
        return fail(PL_ERR_INVALID, "past_cp have to be None or the sequence length has to be an even number");
    if (past_len > h->T) return fail(PL_ERR_INVALID, "pl_set_past_cp: past_len exceeds n_frames");
    DeviceGuard guard(h->cfg.device);
    const int64_t n = (int64_t)(per_utterance ? h->B : 1) * past_len * h->C;
    launch_f32_to_f64(h->stream, past_cp, h->past, n);
    int rc = check_launch();
    if (rc) return rc;
    PL_HIP(hipStreamSynchronize(h->stream));
    if (h->past_len != past_len || h->past_per_utt != (per_utterance ? 1 : 0)) drop_graph(h);   // kernel args change
    h->past_len = past_len;
    h->past_per_utt = per_utterance ? 1 : 0;
    return PL_OK;
}

int pl_reset_optimizer(pl_handle* h) {
    if (!h) return fail(PL_ERR_INVALID, "pl_reset_optimizer: NULL handle");
    DeviceGuard guard(h->cfg.device);
    const size_t nb = sizeof(double) * h->B * h->T * h->C;
    PL_HIP(hipMemsetAsync(h->m, 0, nb, h->stream));
    PL_HIP(hipMemsetAsync(h->v, 0, nb, h->stream));
    PL_HIP(hipMemsetAsync(h->counters, 0, sizeof(int) * 4, h->stream));
    return PL_OK;
}

static int check_ready(pl_handle* h, bool need_sem, const char* who) {
    if (!h->pred.ready()) return fail(PL_ERR_STATE, std::string(who) + ": predictive-model weights are not set");
    if (need_sem && !emb_ready(h)) return fail(PL_ERR_STATE, std::string(who) + ": embedder weights are not set");
    if (h->tube_on() && !tube_ready(h)) return fail(PL_ERR_STATE, std::string(who) + ": weights of the somatosensory models are not set");
    if (!h->have_cp) return fail(PL_ERR_STATE, std::string(who) + ": pl_set_cp has not been called");
    return PL_OK;
}

int pl_step(pl_handle* h, int n_iters, float* loss_log, float* grad_out) {
    if (!h) return fail(PL_ERR_INVALID, "pl_step: NULL handle");
    if (n_iters < 0) return fail(PL_ERR_INVALID, "pl_step: n_iters < 0");
    const bool with_sem = h->need_emb_in_step();
    int rc = check_ready(h, with_sem, "pl_step");
    if (rc) return rc;
    if (!h->have_targets) return fail(PL_ERR_STATE, "pl_step: pl_set_targets has not been called");
    if (with_sem && !h->have_sem_target) return fail(PL_ERR_STATE, "pl_step: objective needs a target_semvec");
    DeviceGuard guard(h->cfg.device);
    SweepChain chain(h->cfg.device, h->stream);
    if (h->cfg.use_graph && !h->graph_exec && n_iters > 0) {
        rc = build_graph(h);
        if (rc) return rc;
    }
    int done = 0;
    while (done < n_iters) {
        const int chunk = (n_iters - done) < h->loss_cap ? (n_iters - done) : h->loss_cap;
        PL_HIP(hipMemsetAsync(h->counters + 1, 0, sizeof(int), h->stream));
        for (int i = 0; i < chunk; ++i) {
            if (h->graph_exec) {
                DBG_G("launch");
                PL_HIP(hipGraphLaunch(h->graph_exec, h->stream));
                DBG_G("launch done");
            }
            else
                enqueue_iteration(h, h->stream);
        }
        if (loss_log)
            PL_HIP(hipMemcpyAsync(loss_log + (size_t)done * h->B * PL_LOSS_COLS, h->loss_rows,
                                  sizeof(float) * chunk * h->B * PL_LOSS_COLS, hipMemcpyDeviceToDevice, h->stream));
        done += chunk;
    }
    if (grad_out && n_iters > 0) launch_f64_to_f32(h->stream, h->grad, grad_out, (int64_t)h->B * h->T * h->C);
    return check_launch();
}

int pl_synchronize(pl_handle* h) {
    if (!h) return fail(PL_ERR_INVALID, "pl_synchronize: NULL handle");
    DeviceGuard guard(h->cfg.device);
    PL_HIP(hipStreamSynchronize(h->stream));
#ifdef PL_STAMPS
    if (const char* path = std::getenv("PL_STAMP_FILE")) {   // phase ticks of the LAST forward / backward sweep
        std::vector<unsigned long long> host(2 * 256 * 8);
        PL_HIP(hipMemcpy(host.data(), h->sweep_stamps, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE* f = std::fopen((std::string(path) + ".sweep").c_str(), "wb")) {
            std::fwrite(host.data(), sizeof(unsigned long long), host.size(), f);
            std::fclose(f);
        }
    }
#endif
    if (h->debug_fused && h->fused_rows16) {   // PAULE_HIP_DEBUG=fused: where the 16-row roles' workgroups sat in the last iteration
        for (int bwd = 0; bwd < 2; ++bwd)
            for (int r = 0; r < h->fused_n_roles; ++r) {
                int ids[32] = {};
                const int* xt = fused_slice(h, r, bwd != 0) + (size_t)((h->Bp + 7) / 8) * h->T * h->flag_stride;
                PL_HIP(hipMemcpyAsync(ids, xt, sizeof(ids), hipMemcpyDeviceToHost, h->stream));
                PL_HIP(hipStreamSynchronize(h->stream));
                if (!ids[0]) continue;   // not an LSTM role
                fprintf(stderr, "[pl] fused16 %s role %d on XCDs:", bwd ? "backward" : "forward", r);
                for (int i = 0; i < 32 && ids[i]; ++i) fprintf(stderr, " %d", ids[i] - 1);
                fprintf(stderr, "\n");
            }
    }
    // on the handle's own stream: a copy on the legacy stream would collide with another handle's graph capture
    int st = 0;
    PL_HIP(hipMemcpyAsync(&st, h->sweep_status, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    PL_HIP(hipStreamSynchronize(h->stream));
    if (st == 2) {   // the residency census of a fused launch: not all of its workgroups got a CU
        (void)hipMemsetAsync(h->sweep_status, 0, sizeof(int), h->stream);
        if (h->sweep_xchg_tok) (void)hipMemsetAsync(h->sweep_xchg_tok, 0, h->sweep_xchg_tok_bytes, h->stream);   // the sweeps behind it gave up too
        return fail(PL_ERR_STATE, "fused LSTM launch: its workgroups were not all resident within the census bound -- something else holds "
                                  "CUs of this GPU (one process per GPU, see paule_hip.h); results of the last pl_step are invalid");
    }
    if (st != 0) {
        (void)hipMemsetAsync(h->sweep_status, 0, sizeof(int), h->stream);
        // an abandoned token-form sweep leaves tiles behind that a later launch could take for its own: back to "retired"
        if (h->sweep_xchg_tok) (void)hipMemsetAsync(h->sweep_xchg_tok, 0, h->sweep_xchg_tok_bytes, h->stream);
        return fail(PL_ERR_HIP, "persistent LSTM sweep: a bounded in-kernel wait timed out (workgroups of one batch group were not "
                                "co-resident?); results of the last pl_step are invalid");
    }
    return PL_OK;
}

int pl_get_cp(pl_handle* h, float* cp_out) {
    if (!h || !cp_out) return fail(PL_ERR_INVALID, "pl_get_cp: NULL argument");
    DeviceGuard guard(h->cfg.device);
    launch_f64_to_f32(h->stream, h->x, cp_out, (int64_t)h->B * h->T * h->C);
    return check_launch();
}

int pl_get_pred(pl_handle* h, float* pred_mel_out, float* pred_semvec_out) {
    if (!h) return fail(PL_ERR_INVALID, "pl_get_pred: NULL handle");
    int rc = check_ready(h, pred_semvec_out != nullptr, "pl_get_pred");
    if (rc) return rc;
    if (pred_semvec_out && h->emb.L == 0) return fail(PL_ERR_INVALID, "pl_get_pred: pred_semvec requested but the handle has no embedder");
    DeviceGuard guard(h->cfg.device);
    SweepChain chain(h->cfg.device, h->stream);
    hipStream_t st = h->stream;
    pred_forward(h, st);
    if (pred_mel_out)
        PL_HIP(hipMemcpyAsync(pred_mel_out, h->mel_bm, sizeof(float) * h->B * h->Tp * h->M, hipMemcpyDeviceToDevice, st));
    if (pred_semvec_out) {
        emb_forward(h, st, nullptr, h->mel_bm);
        launch_unpad_rows(st, h->sem, h->B, h->S, h->Sp, pred_semvec_out);
    }
    return check_launch();
}

// the predictive model WITHOUT the half sequence: post_linear(lstm(cp)) for every frame, [B, n_frames, mel_dim]
// (ForwardModel(apply_half_sequence=False).forward, paule/models.py:348-356; how the reference builds cp_tube_model, paule/paule.py:232-237)
int pl_get_pred_frames(pl_handle* h, float* frames_out) {
    if (!h || !frames_out) return fail(PL_ERR_INVALID, "pl_get_pred_frames: NULL argument");
    int rc = check_ready(h, false, "pl_get_pred_frames");
    if (rc) return rc;
    DeviceGuard guard(h->cfg.device);
    SweepChain chain(h->cfg.device, h->stream);
    pred_forward(h, h->stream);
    launch_tm_to_bm(h->stream, h->Y, h->B, h->T, h->M, h->Bp, h->Mp, frames_out);
    return check_launch();
}

int pl_get_tube_pred(pl_handle* h, float* pred_tube_out, float* pred_tube_mel_out, float* pred_tube_semvec_out) {
    if (!h) return fail(PL_ERR_INVALID, "pl_get_tube_pred: NULL handle");
    if (!h->tube_on()) return fail(PL_ERR_INVALID, "pl_get_tube_pred: the handle has no somatosensory models (pl_config.cp_tube_layers)");
    int rc = check_ready(h, false, "pl_get_tube_pred");
    if (rc) return rc;
    DeviceGuard guard(h->cfg.device);
    SweepChain chain(h->cfg.device, h->stream);
    hipStream_t st = h->stream;
    launch_pack_cp(st, h->dt, h->x, h->B, h->T, h->C, h->X0, h->Bp, h->Cp);
    tube_forward(h, st);
    if (pred_tube_out) {   // time-major activation [T][Bp][Up] -> [B][T][U] f32 (Y2 is free again: reuse it as the f32 staging)
        launch_act_to_f32(st, h->dt, h->tube_tm, h->Y2, (int64_t)h->T * h->Bp * h->Up);
        launch_tm_to_bm(st, h->Y2, h->B, h->T, h->U, h->Bp, h->Up, pred_tube_out);
    }
    if (pred_tube_mel_out)
        PL_HIP(hipMemcpyAsync(pred_tube_mel_out, h->mel2_bm, sizeof(float) * h->B * h->Tp * h->M, hipMemcpyDeviceToDevice, st));
    if (pred_tube_semvec_out) launch_unpad_rows(st, h->sem2, h->B, h->S, h->Sp, pred_tube_semvec_out);
    return check_launch();
}

int pl_embed_tube(pl_handle* h, const float* tube, float* tube_mel_out, float* tube_semvec_out) {
    if (!h || !tube) return fail(PL_ERR_INVALID, "pl_embed_tube: NULL argument");
    if (!h->tube_on()) return fail(PL_ERR_INVALID, "pl_embed_tube: the handle has no somatosensory models (pl_config.cp_tube_layers)");
    if (!tube_ready(h)) return fail(PL_ERR_STATE, "pl_embed_tube: weights of the somatosensory models are not set");
    DeviceGuard guard(h->cfg.device);
    SweepChain chain(h->cfg.device, h->stream);
    hipStream_t st = h->stream;
    launch_pack_mel(st, h->dt, tube, h->B, h->T, h->U, h->tube_tm, h->Bp, h->Up);
    tube_heads_forward(h, st);
    if (tube_mel_out)
        PL_HIP(hipMemcpyAsync(tube_mel_out, h->mel2_bm, sizeof(float) * h->B * h->Tp * h->M, hipMemcpyDeviceToDevice, st));
    if (tube_semvec_out) launch_unpad_rows(st, h->sem2, h->B, h->S, h->Sp, tube_semvec_out);
    return check_launch();
}

int pl_embed_mel(pl_handle* h, const float* mel, const int32_t* lens, float* semvec_out) {
    if (!h || !mel || !semvec_out) return fail(PL_ERR_INVALID, "pl_embed_mel: NULL argument");
    if (h->emb.L == 0) return fail(PL_ERR_INVALID, "pl_embed_mel: the handle has no embedder");
    if (!emb_ready(h)) return fail(PL_ERR_STATE, "pl_embed_mel: embedder weights are not set");
    DeviceGuard guard(h->cfg.device);
    SweepChain chain(h->cfg.device, h->stream);
    hipStream_t st = h->stream;
    launch_pack_mel(st, h->dt, mel, h->B, h->Tp, h->M, h->mel_tm, h->Bp, h->Mp);
    emb_forward(h, st, lens, mel);
    launch_unpad_rows(st, h->sem, h->B, h->S, h->Sp, semvec_out);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------
// continued learning of the predictive model (paule/paule.py:1353-1379)
// ---------------------------------------------------------------------------------------------------
namespace {

int ensure_train_state(pl_handle* h, Model& md) {
    if (md.train_ready) return PL_OK;
    int rc;
    for (LstmLayer& ly : md.layers) {
        for (ParamState* ps : {&ly.p_wih, &ly.p_whh, &ly.p_bih, &ly.p_bhh}) {
            if ((rc = dev_alloc(h, &ps->m, ps->n))) return rc;
            if ((rc = dev_alloc(h, &ps->v, ps->n))) return rc;
        }
        if ((rc = dev_alloc(h, &ly.gWih, (size_t)4 * md.Hp * ly.in_p))) return rc;
        if ((rc = dev_alloc(h, &ly.gWhh, (size_t)4 * md.Hp * md.Hp))) return rc;
        if ((rc = dev_alloc(h, &ly.gb, (size_t)4 * md.Hp))) return rc;
    }
    for (ParamState* ps : {&md.p_wlin, &md.p_blin}) {
        if ((rc = dev_alloc(h, &ps->m, ps->n))) return rc;
        if ((rc = dev_alloc(h, &ps->v, ps->n))) return rc;
    }
    if ((rc = dev_alloc(h, &md.gWlin, (size_t)md.out_p * md.Hp))) return rc;
    if ((rc = dev_alloc(h, &md.gblin, (size_t)md.out_p))) return rc;
    // split-K partials: two splits of the largest product (dW_hh), which also covers many splits of the small ones
    md.train_scratch_bytes = (size_t)2 * 4 * md.Hp * md.Hp * sizeof(float);
    if (md.train_scratch_bytes < train_scratch_bytes(4 * md.Hp, 128)) md.train_scratch_bytes = train_scratch_bytes(4 * md.Hp, 128);
    if ((rc = dev_alloc(h, &md.train_scratch, md.train_scratch_bytes / sizeof(float)))) return rc;
    if ((rc = dev_alloc(h, &md.colsum_part, (size_t)64 * 4 * md.Hp))) return rc;
    md.train_ready = true;
    return PL_OK;
}

}  // namespace

// One optimizer step of a ForwardModel of the handle on a mini-batch (paule/paule.py:1372-1377 for pred_model, :1386-1404 for
// the cp -> tube and tube -> mel models of the somatosensory path): forward, batch RMSE, backward with weight gradients, Adam.
int pl_train_model_step(pl_handle* h, int model_id, int n_rows, int n_frames, const float* input, const float* target, float lr,
                        float beta1, float beta2, float eps, float* loss_out) {
    if (!h || !input || !target) return fail(PL_ERR_INVALID, "pl_train_model_step: NULL argument");
    if (model_id != PL_MODEL_PRED && model_id != PL_MODEL_CP_TUBE && model_id != PL_MODEL_TUBE_MEL)
        return fail(PL_ERR_INVALID, "pl_train_model_step: model_id has to be PL_MODEL_PRED, PL_MODEL_CP_TUBE or PL_MODEL_TUBE_MEL");
    if (n_rows < 1 || n_rows > h->Bp) return fail(PL_ERR_INVALID, "pl_train_model_step: n_rows has to be in [1, batch rounded up to 16]");
    if (n_frames < 2 || n_frames > h->T) return fail(PL_ERR_INVALID, "pl_train_model_step: n_frames has to be in [2, n_frames of the handle]");
    if (!(lr > 0.f) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f))
        return fail(PL_ERR_INVALID, "pl_train_model_step: bad optimizer hyper-parameter");
    Model& p = *model_by_id(h, model_id);
    if (p.L == 0) return fail(PL_ERR_INVALID, "pl_train_model_step: the handle has no such model");
    if (!p.ready()) return fail(PL_ERR_STATE, "pl_train_model_step: the model's weights are not set");
    DeviceGuard guard(h->cfg.device);
    SweepChain chain(h->cfg.device, h->stream);
    int rc = ensure_train_state(h, p);
    if (rc) return rc;
    hipStream_t st = h->stream;
    struct RowsGuard {   // training sums weight gradients over the 16-row block of the mini-batch and relies on EXACT zeros in the
        pl_handle* h; int prev;   // padded rows: the few-row kernels leave those rows of the exchange untouched, so they stay off here
        explicit RowsGuard(pl_handle* hh) : h(hh), prev(hh->rows_in_use) { h->rows_in_use = 0; }
        ~RowsGuard() { h->rows_in_use = prev; }
    } rows_guard(h);
    const bool pooled = model_id != PL_MODEL_CP_TUBE;   // ForwardModel(apply_half_sequence=...) (paule/paule.py:236, :252)
    const int Bp = h->Bp, T = n_frames, To = pooled ? n_frames / 2 : n_frames, nb = pad16(n_rows);
    // buffers by model: time-major input slab, f32 post_linear output, batch-major prediction, output gradient
    void* in_tm = model_id == PL_MODEL_TUBE_MEL ? h->tube_tm : h->X0;
    float* Yf = model_id == PL_MODEL_PRED ? h->Y : h->Y2;
    float* out_bm = model_id == PL_MODEL_PRED ? h->mel_bm : h->mel2_bm;
    void* dYb = model_id == PL_MODEL_PRED ? h->dY : model_id == PL_MODEL_TUBE_MEL ? h->dY2 : h->dYt;
    // forward: Y_hat = model(batch_input) (paule/paule.py:1372, :1387, :1397); rows >= n_rows of every slab are zero input
    launch_pack_mel(st, h->dt, input, n_rows, T, p.in, in_tm, Bp, p.in_p);
    model_forward(h, st, p, in_tm, T);
    const LstmLayer& top = p.layers[p.L - 1];
    launch_gemm_nt(st, h->dt, true, top.h, p.Hp, p.Wlin, p.Hp, p.blin, Yf, p.out_p, T * Bp, p.out_p, p.Hp);
    if (pooled)
        launch_pool_mel(st, h->dt, Yf, n_rows, T, p.out, Bp, p.out_p, out_bm, model_id == PL_MODEL_PRED ? h->mel_tm : h->mel2_tm);
    else
        launch_tm_to_bm(st, Yf, n_rows, T, p.out, Bp, p.out_p, out_bm);
    // loss = rmse over the whole batch (RMSELoss(eps=0), paule/paule.py:288, :301, :306, :1375)
    launch_train_rmse(st, out_bm, target, (int64_t)n_rows * To * p.out, h->scal, loss_out);
    // backward (paule/paule.py:1376): dY, post_linear gradients, then the recurrences with weight gradients
    launch_train_dy(st, h->dt, out_bm, target, h->scal, n_rows, T, To, p.out, Bp, p.out_p, dYb, pooled);
    launch_gemm_tn(st, h->dt, dYb, p.out_p, top.h, p.Hp, p.gWlin, p.Hp, p.out_p, p.Hp, Bp, nb, T, 0, 0, p.train_scratch,
                   p.train_scratch_bytes, h->n_cu, h->tn_bf16);
    launch_colsum(st, h->dt, dYb, p.out_p, p.out_p, Bp, nb, T, 0, p.gblin, p.colsum_part);
    launch_gemm_nt(st, h->dt, false, dYb, p.out_p, p.WlinT, p.out_p, nullptr, p.dh_ext, p.Hp, T * Bp, p.Hp, p.out_p);
    model_backward(h, st, p, nullptr, nullptr, nb, in_tm, T);
    // optimizer.step() (paule/paule.py:1377; torch.optim.Adam defaults, :287, :300, :305) + refresh of the packed compute copies
    p.train_steps += 1;
    AdamHyper hp{(double)lr, (double)beta1, (double)beta2, (double)eps, 1.0 - std::pow((double)beta1, (double)p.train_steps),
                 1.0 - std::pow((double)beta2, (double)p.train_steps)};
    for (LstmLayer& ly : p.layers) {
        launch_adam_matrix(st, h->dt, ly.gWih, 4, p.H, ly.in, p.Hp, ly.in_p, ly.p_wih.x, ly.p_wih.m, ly.p_wih.v, ly.Wih, ly.WihT, hp);
        launch_adam_matrix(st, h->dt, ly.gWhh, 4, p.H, p.H, p.Hp, p.Hp, ly.p_whh.x, ly.p_whh.m, ly.p_whh.v, ly.Whh, ly.WhhT, hp);
        launch_adam_bias(st, ly.gb, 4, p.H, p.Hp, ly.p_bih.x, ly.p_bih.m, ly.p_bih.v, ly.p_bhh.x, ly.p_bhh.m, ly.p_bhh.v, ly.bias, hp);
    }
    launch_adam_matrix(st, h->dt, p.gWlin, 1, p.out, p.H, p.out_p, p.Hp, p.p_wlin.x, p.p_wlin.m, p.p_wlin.v, p.Wlin, p.WlinT, hp);
    launch_adam_bias(st, p.gblin, 1, p.out, p.out_p, p.p_blin.x, p.p_blin.m, p.p_blin.v, nullptr, nullptr, nullptr, p.blin, hp);
    return check_launch();
}

int pl_train_pred_step(pl_handle* h, int n_rows, int n_frames, const float* cp, const float* mel_target, float lr, float beta1,
                       float beta2, float eps, float* loss_out) {
    return pl_train_model_step(h, PL_MODEL_PRED, n_rows, n_frames, cp, mel_target, lr, beta1, beta2, eps, loss_out);
}

namespace {
bool trainable_id(pl_handle* h, int model_id) {
    return (model_id == PL_MODEL_PRED || model_id == PL_MODEL_CP_TUBE || model_id == PL_MODEL_TUBE_MEL) && model_by_id(h, model_id)->L > 0;
}
}  // namespace

int pl_reset_pred_optimizer(pl_handle* h) { return pl_reset_model_optimizer(h, PL_MODEL_PRED); }

int pl_reset_model_optimizer(pl_handle* h, int model_id) {
    if (!h) return fail(PL_ERR_INVALID, "pl_reset_model_optimizer: NULL handle");
    if (!trainable_id(h, model_id)) return fail(PL_ERR_INVALID, "pl_reset_model_optimizer: no such trainable model in this handle");
    DeviceGuard guard(h->cfg.device);
    Model& p = *model_by_id(h, model_id);
    p.train_steps = 0;
    if (!p.train_ready) return PL_OK;
    auto zero = [&](ParamState& ps) -> int {
        PL_HIP(hipMemsetAsync(ps.m, 0, sizeof(double) * ps.n, h->stream));
        PL_HIP(hipMemsetAsync(ps.v, 0, sizeof(double) * ps.n, h->stream));
        return PL_OK;
    };
    int rc;
    for (LstmLayer& ly : p.layers)
        for (ParamState* ps : {&ly.p_wih, &ly.p_whh, &ly.p_bih, &ly.p_bhh})
            if ((rc = zero(*ps))) return rc;
    if ((rc = zero(p.p_wlin))) return rc;
    return zero(p.p_blin);
}

// Adam state of the predictive model's optimiser (torch.optim.Adam state_dict: step, exp_avg, exp_avg_sq per parameter), so
// that the reference's `pred_optimizer` -- which lives as long as the Paule instance (paule/paule.py:284-287) and is saved /
// restored by its users (docs/examples/minimal_example.py:51, continue_planning.py:27) -- can outlive a handle.
// which: 1 exp_avg, 2 exp_avg_sq.  layer >= 0: the four LSTM tensors of that layer; layer = -1: post_linear (w_ih = weight,
// b_ih = bias; w_hh, b_hh ignored).  float32 device pointers in torch layout.
namespace {
int pred_state_io(pl_handle* h, int model_id, bool set, int layer, int which, float* w_ih, float* w_hh, float* b_ih, float* b_hh,
                  const char* who) {
    if (!h || !w_ih || !b_ih || (layer >= 0 && (!w_hh || !b_hh))) return fail(PL_ERR_INVALID, std::string(who) + ": NULL argument");
    if (which != 1 && which != 2) return fail(PL_ERR_INVALID, std::string(who) + ": which has to be 1 (exp_avg) or 2 (exp_avg_sq)");
    if (!trainable_id(h, model_id)) return fail(PL_ERR_INVALID, std::string(who) + ": no such trainable model in this handle");
    Model& p = *model_by_id(h, model_id);
    if (layer < -1 || layer >= p.L) return fail(PL_ERR_INVALID, std::string(who) + ": layer out of range");
    DeviceGuard guard(h->cfg.device);
    int rc = ensure_train_state(h, p);
    if (rc) return rc;
    auto io = [&](ParamState& ps, float* ptr) {
        double* d = which == 1 ? ps.m : ps.v;
        if (set) launch_f32_to_f64(h->stream, ptr, d, (int64_t)ps.n);
        else launch_f64_to_f32(h->stream, d, ptr, (int64_t)ps.n);
    };
    if (layer >= 0) {
        LstmLayer& ly = p.layers[layer];
        io(ly.p_wih, w_ih); io(ly.p_whh, w_hh); io(ly.p_bih, b_ih); io(ly.p_bhh, b_hh);
    } else {
        io(p.p_wlin, w_ih); io(p.p_blin, b_ih);
    }
    rc = check_launch();
    if (rc) return rc;
    PL_HIP(hipStreamSynchronize(h->stream));
    return PL_OK;
}
}  // namespace

int pl_get_pred_optimizer_state(pl_handle* h, int layer, int which, float* w_ih, float* w_hh, float* b_ih, float* b_hh) {
    return pred_state_io(h, PL_MODEL_PRED, false, layer, which, w_ih, w_hh, b_ih, b_hh, "pl_get_pred_optimizer_state");
}
int pl_set_pred_optimizer_state(pl_handle* h, int layer, int which, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh) {
    return pred_state_io(h, PL_MODEL_PRED, true, layer, which, const_cast<float*>(w_ih), const_cast<float*>(w_hh), const_cast<float*>(b_ih),
                         const_cast<float*>(b_hh), "pl_set_pred_optimizer_state");
}
int64_t pl_get_pred_optimizer_step(const pl_handle* h) { return h ? (int64_t)h->pred.train_steps : -1; }
int pl_set_pred_optimizer_step(pl_handle* h, int64_t step) {
    if (!h || step < 0) return fail(PL_ERR_INVALID, "pl_set_pred_optimizer_step: bad argument");
    h->pred.train_steps = (long long)step;
    return PL_OK;
}
// the same for any trainable model of the handle (tube_optimizer / tube_mel_optimizer, paule/paule.py:296-306)
int pl_get_model_optimizer_state(pl_handle* h, int model_id, int layer, int which, float* w_ih, float* w_hh, float* b_ih, float* b_hh) {
    return pred_state_io(h, model_id, false, layer, which, w_ih, w_hh, b_ih, b_hh, "pl_get_model_optimizer_state");
}
int pl_set_model_optimizer_state(pl_handle* h, int model_id, int layer, int which, const float* w_ih, const float* w_hh, const float* b_ih,
                                 const float* b_hh) {
    return pred_state_io(h, model_id, true, layer, which, const_cast<float*>(w_ih), const_cast<float*>(w_hh), const_cast<float*>(b_ih),
                         const_cast<float*>(b_hh), "pl_set_model_optimizer_state");
}
int64_t pl_get_model_optimizer_step(pl_handle* h, int model_id) {
    return (h && trainable_id(h, model_id)) ? (int64_t)model_by_id(h, model_id)->train_steps : -1;
}
int pl_set_model_optimizer_step(pl_handle* h, int model_id, int64_t step) {
    if (!h || step < 0 || !trainable_id(h, model_id)) return fail(PL_ERR_INVALID, "pl_set_model_optimizer_step: bad argument");
    model_by_id(h, model_id)->train_steps = (long long)step;
    return PL_OK;
}

int pl_get_lstm_weights(pl_handle* h, int model_id, int layer, float* w_ih, float* w_hh, float* b_ih, float* b_hh) {
    if (!h || !w_ih || !w_hh || !b_ih || !b_hh) return fail(PL_ERR_INVALID, "pl_get_lstm_weights: NULL argument");
    if (model_id < PL_MODEL_PRED || model_id > PL_MODEL_TUBE_EMBED) return fail(PL_ERR_INVALID, "pl_get_lstm_weights: bad model_id");
    Model& md = *model_by_id(h, model_id);
    if (layer < 0 || layer >= md.L) return fail(PL_ERR_INVALID, "pl_get_lstm_weights: layer out of range for this model");
    LstmLayer& ly = md.layers[layer];
    if (!ly.set) return fail(PL_ERR_STATE, "pl_get_lstm_weights: the layer's weights are not set");
    DeviceGuard guard(h->cfg.device);
    launch_f64_to_f32(h->stream, ly.p_wih.x, w_ih, (int64_t)ly.p_wih.n);
    launch_f64_to_f32(h->stream, ly.p_whh.x, w_hh, (int64_t)ly.p_whh.n);
    launch_f64_to_f32(h->stream, ly.p_bih.x, b_ih, (int64_t)ly.p_bih.n);
    launch_f64_to_f32(h->stream, ly.p_bhh.x, b_hh, (int64_t)ly.p_bhh.n);
    int rc = check_launch();
    if (rc) return rc;
    PL_HIP(hipStreamSynchronize(h->stream));
    return PL_OK;
}

int pl_get_linear(pl_handle* h, int model_id, float* w, float* b) {
    if (!h || !w || !b) return fail(PL_ERR_INVALID, "pl_get_linear: NULL argument");
    if (model_id < PL_MODEL_PRED || model_id > PL_MODEL_TUBE_EMBED) return fail(PL_ERR_INVALID, "pl_get_linear: bad model_id");
    Model& md = *model_by_id(h, model_id);
    if (md.L == 0 || !md.lin_set) return fail(PL_ERR_STATE, "pl_get_linear: the output layer's weights are not set");
    DeviceGuard guard(h->cfg.device);
    launch_f64_to_f32(h->stream, md.p_wlin.x, w, (int64_t)md.p_wlin.n);
    launch_f64_to_f32(h->stream, md.p_blin.x, b, (int64_t)md.p_blin.n);
    int rc = check_launch();
    if (rc) return rc;
    PL_HIP(hipStreamSynchronize(h->stream));
    return PL_OK;
}

// ---------------------------------------------------------------------------------------------------
// embedder variants (paule/models.py:362-409, :432-446)
// ---------------------------------------------------------------------------------------------------
int pl_set_embedder_output(pl_handle* h, const float* w, const float* b) {
    if (!h || !w || !b) return fail(PL_ERR_INVALID, "pl_set_embedder_output: NULL argument");
    if (h->emb.L == 0 || h->emb_post == 0)
        return fail(PL_ERR_INVALID, "pl_set_embedder_output: the handle's embedder has no post_linear head (pl_config.emb_post_size)");
    DeviceGuard guard(h->cfg.device);
    hipStream_t st = h->stream;
    launch_pack_matrix(st, h->dt, w, 1, h->S, h->emb_post, h->Wup, h->Sp, h->emb_post_p, false);
    launch_pack_matrix(st, h->dt, w, 1, h->S, h->emb_post, h->WupT, h->Sp, h->emb_post_p, true);
    launch_pack_bias(st, b, nullptr, 1, h->S, h->bup, h->Sp);
    h->up_set = true;
    int rc = check_launch();
    if (rc) return rc;
    PL_HIP(hipStreamSynchronize(st));
    return PL_OK;
}

int pl_set_embedder_conv(pl_handle* h, int block, int idx, const float* w, const float* b) {
    if (!h || !w || !b) return fail(PL_ERR_INVALID, "pl_set_embedder_conv: NULL argument");
    if (h->emb_blocks == 0) return fail(PL_ERR_INVALID, "pl_set_embedder_conv: the handle's embedder has no mel blocks (pl_config.emb_mel_blocks)");
    if (block < 0 || block >= h->emb_blocks || idx < 0 || idx >= 3) return fail(PL_ERR_INVALID, "pl_set_embedder_conv: no such convolution");
    const size_t G = h->M / 3, blk = 3 * G * 16;
    DeviceGuard guard(h->cfg.device);
    PL_HIP(hipMemcpyAsync(h->emb_conv + block * blk + idx * G * 15, w, sizeof(float) * G * 15, hipMemcpyDeviceToDevice, h->stream));
    PL_HIP(hipMemcpyAsync(h->emb_conv + block * blk + 3 * G * 15 + idx * G, b, sizeof(float) * G, hipMemcpyDeviceToDevice, h->stream));
    PL_HIP(hipStreamSynchronize(h->stream));
    h->emb_conv_set[(size_t)block * 3 + idx] = 1;
    return PL_OK;
}

// ---------------------------------------------------------------------------------------------------
// inverse model forward (paule/models.py:210-247; paule/paule.py:550-556)
// ---------------------------------------------------------------------------------------------------
int pl_set_inverse_conv(pl_handle* h, int kind, int block, int idx, const float* w, const float* b) {
    if (!h || !w || !b) return fail(PL_ERR_INVALID, "pl_set_inverse_conv: NULL argument");
    if (h->inv.L == 0) return fail(PL_ERR_INVALID, "pl_set_inverse_conv: the handle has no inverse model (pl_config.inv_layers)");
    const size_t G = h->M / 3, C = h->C;
    const size_t mel_block = 3 * G * 16, res_base = (size_t)h->inv_mel_blocks * mel_block, rw_base = res_base + (size_t)h->inv_res_blocks * 2 * C * 6;
    float *dw = nullptr, *db = nullptr;
    size_t nw = 0, nbias = 0, slot = 0;
    if (kind == PL_CONV_MEL && block >= 0 && block < h->inv_mel_blocks && idx >= 0 && idx < 3) {
        dw = h->inv_conv + block * mel_block + idx * G * 15;
        db = h->inv_conv + block * mel_block + 3 * G * 15 + idx * G;
        nw = G * 15; nbias = G; slot = (size_t)block * 3 + idx;
    } else if (kind == PL_CONV_RES && block >= 0 && block < h->inv_res_blocks && idx >= 0 && idx < 2) {
        dw = h->inv_conv + res_base + ((size_t)block * 2 + idx) * C * 6;
        db = dw + C * 5;
        nw = C * 5; nbias = C; slot = (size_t)h->inv_mel_blocks * 3 + (size_t)block * 2 + idx;
    } else if (kind == PL_CONV_RW && block == 0 && idx == 0) {
        dw = h->inv_conv + rw_base;
        db = dw + C * 10;
        nw = C * 10; nbias = C; slot = h->inv_conv_set.size() - 1;
    } else {
        return fail(PL_ERR_INVALID, "pl_set_inverse_conv: no such convolution in this inverse model");
    }
    DeviceGuard guard(h->cfg.device);
    PL_HIP(hipMemcpyAsync(dw, w, sizeof(float) * nw, hipMemcpyDeviceToDevice, h->stream));
    PL_HIP(hipMemcpyAsync(db, b, sizeof(float) * nbias, hipMemcpyDeviceToDevice, h->stream));
    PL_HIP(hipStreamSynchronize(h->stream));
    h->inv_conv_set[slot] = 1;
    return PL_OK;
}

int pl_inverse_forward(pl_handle* h, const float* mel, int n_mel_frames, float* cp_out, int clip) {
    if (!h || !mel || !cp_out) return fail(PL_ERR_INVALID, "pl_inverse_forward: NULL argument");
    Model& iv = h->inv;
    if (iv.L == 0) return fail(PL_ERR_INVALID, "pl_inverse_forward: the handle has no inverse model (pl_config.inv_layers)");
    if (n_mel_frames < 1 || n_mel_frames > h->Tp) return fail(PL_ERR_INVALID, "pl_inverse_forward: n_mel_frames has to be in [1, n_frames / 2]");
    if (!iv.ready()) return fail(PL_ERR_STATE, "pl_inverse_forward: inverse-model LSTM / post_linear weights are not set");
    for (size_t i = 0; i < h->inv_conv_set.size(); ++i)
        if (!h->inv_conv_set[i] && !(i + 1 == h->inv_conv_set.size() && h->inv_res_blocks == 0))
            return fail(PL_ERR_STATE, "pl_inverse_forward: a convolution of the inverse model is not set");
    DeviceGuard guard(h->cfg.device);
    SweepChain chain(h->cfg.device, h->stream);
    hipStream_t st = h->stream;
    const int B = h->B, Bp = h->Bp, M = h->M, C = h->C, Tn = n_mel_frames, T2 = 2 * n_mel_frames;
    const size_t G = M / 3;
    const size_t mel_block = 3 * G * 16, res_base = (size_t)h->inv_mel_blocks * mel_block, rw_base = res_base + (size_t)h->inv_res_blocks * 2 * C * 6;
    // mel smoothing with residual connections (paule/models.py:220-228)
    const float* x = mel;
    for (int i = 0; i < h->inv_mel_blocks; ++i) {
        const float* w = h->inv_conv + i * mel_block;
        launch_mel_block(st, x, B, Tn, M, w, w + 3 * G * 15, h->inv_x[i & 1]);
        x = h->inv_x[i & 1];
    }
    // velocity / acceleration features, LSTM stack, post_linear (:230-232)
    launch_vel_acc_pack(st, h->dt, x, B, Tn, M, h->inv_in, Bp, iv.in_p);
    model_forward(h, st, iv, h->inv_in, Tn);
    launch_gemm_nt(st, h->dt, true, iv.layers[iv.L - 1].h, iv.Hp, iv.Wlin, iv.Hp, iv.blin, h->inv_Y, h->Cp, Tn * Bp, h->Cp, iv.Hp);
    // double_sequence (:233), time smoothing with residual connections (:235-238), resid_weighting (:240-243)
    launch_double_seq(st, h->inv_Y, B, Tn, C, Bp, h->Cp, h->inv_z[0]);
    if (h->inv_res_blocks == 0) {
        launch_clip_copy(st, h->inv_z[0], (int64_t)B * T2 * C, clip, cp_out);
        return check_launch();
    }
    const float* cur = h->inv_z[0];
    for (int i = 0; i < h->inv_res_blocks; ++i) {
        const float* w1 = h->inv_conv + res_base + ((size_t)i * 2) * C * 6;
        const float* w2 = w1 + C * 6;
        launch_time_conv5(st, cur, B, T2, C, w1, w1 + C * 5, nullptr, h->inv_z[2]);
        float* nxt = (cur == h->inv_z[1]) ? const_cast<float*>(cur) : h->inv_z[1];   // z[1] <- conv2(z[2]) + cur (elementwise in place is safe)
        launch_time_conv5(st, h->inv_z[2], B, T2, C, w2, w2 + C * 5, cur, nxt);
        cur = nxt;
    }
    const float* rw = h->inv_conv + rw_base;
    launch_resid_weight(st, cur, h->inv_z[0], B, T2, C, rw, rw + C * 10, clip, cp_out);
    return check_launch();
}

int pl_bench_kernel(pl_handle* h, int kernel, int model_id, int reps, float* avg_ms_out, double* flops_per_launch_out) {
    if (!h || !avg_ms_out || reps < 1) return fail(PL_ERR_INVALID, "pl_bench_kernel: bad argument");
    if (kernel < PL_KERNEL_LSTM_FWD_STEP || kernel > PL_KERNEL_FUSED_BWD) return fail(PL_ERR_INVALID, "pl_bench_kernel: unknown kernel");
    if (kernel == PL_KERNEL_FUSED_BWD) {
        if (!h->fused_bwd_ok || !h->pred.ready() || !h->emb.ready())
            return fail(PL_ERR_UNSUPPORTED, "pl_bench_kernel: this handle does not run the fused backward launch");
        DeviceGuard guard(h->cfg.device);
        SweepChain chain(h->cfg.device, h->stream);
        hipEvent_t e0, e1;
        PL_HIP(hipEventCreate(&e0));
        PL_HIP(hipEventCreate(&e1));
        PL_HIP(hipEventRecord(e0, h->stream));
        for (int i = 0; i < reps; ++i) {   // the launch alone (+ the flag zeroing), on whatever the last iteration left in the stashes: its
            zero_all_sweep_slots(h, h->stream);   // time does not depend on the data, and every pl_step rebuilds what it overwrites
            FusedArgs a{};
            fused_common_args(h, a, h->fused_grid_bwd, h->fused_tab_bwd, h->fused_roles_bwd, true);
            if (h->fused_rows16) launch_fused_bwd16(h->stream, h->pred.Hp, h->emb.Hp, a);
            else launch_fused_bwd(h->stream, h->pred.Hp, h->emb.Hp, a);
            h->sweep_slot = -1;
        }
        PL_HIP(hipEventRecord(e1, h->stream));
        PL_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        PL_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *avg_ms_out = ms / reps;
        if (flops_per_launch_out) {   // backward-data recurrences of all layers, the dL/dh products between the embedder's layers, the
            const double Hh = h->pred.H, He = h->emb.H;   // embedder's input gradient (-> mel) and the backward mel head
            // predictor: every layer's recurrence and, for a stacked one, the dL/dh product between two layers (one layer was counted)
            double f = h->pred.L * 2.0 * 4 * Hh * Hh * (h->T - 1) + (h->pred.L - 1) * 2.0 * 4 * Hh * Hh * h->T + 2.0 * Hh * h->M * h->Tp;
            f += h->emb.L * 2.0 * 4 * He * He * (h->Tp - 1) + (h->emb.L - 1) * 2.0 * 4 * He * He * h->Tp + 2.0 * 4 * He * h->M * h->Tp;
            *flops_per_launch_out = (double)h->B * f;
        }
        return check_launch();
    }
    if (kernel == PL_KERNEL_FUSED_FWD) {
        if (!h->fused_fwd_ok || !h->pred.ready() || !h->emb.ready())
            return fail(PL_ERR_UNSUPPORTED, "pl_bench_kernel: this handle does not run the fused forward launch");
        DeviceGuard guard(h->cfg.device);
        SweepChain chain(h->cfg.device, h->stream);
        hipEvent_t e0, e1;
        PL_HIP(hipEventCreate(&e0));
        PL_HIP(hipEventCreate(&e1));
        PL_HIP(hipEventRecord(e0, h->stream));
        for (int i = 0; i < reps; ++i) {
            zero_all_sweep_slots(h, h->stream);
            (void)fused_acoustic_forward(h, h->stream);
            h->sweep_slot = -1;
        }
        PL_HIP(hipEventRecord(e1, h->stream));
        PL_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        PL_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *avg_ms_out = ms / reps;
        if (flops_per_launch_out)
            *flops_per_launch_out = (double)h->B * ((lstm_flops_per_step(h->pred.L, h->pred.H, h->C) + 2.0 * h->pred.H * h->M) * h->T +
                                                    lstm_flops_per_step(h->emb.L, h->emb.H, h->M) * h->Tp);
        return check_launch();
    }
    Model& md = model_id == PL_MODEL_EMBED ? h->emb : h->pred;
    if (md.L == 0 || !md.ready()) return fail(PL_ERR_STATE, "pl_bench_kernel: model weights are not set");
    DeviceGuard guard(h->cfg.device);
    SweepChain chain(h->cfg.device, h->stream);
    LstmLayer& ly = md.layers[0];
    const int Bp = h->Bp, Hp = md.Hp, Tl = md.Tl;
    const size_t a = h->act;
    if (kernel >= PL_KERNEL_LSTM_FWD_SWEEP) {
        const bool bwd = kernel == PL_KERNEL_LSTM_BWD_SWEEP;
        const int grid = sweep_grid_for(h, Hp, bwd);
        if (grid <= 0) return fail(PL_ERR_UNSUPPORTED, "pl_bench_kernel: the persistent sweep is not used for this dtype / shape");
        LstmSweepArgs s{};
        s.Bp = Bp; s.T = Tl; s.G = ly.G; s.W = bwd ? ly.WhhT : ly.Whh; s.h = ly.h; s.c = ly.c;
        s.group_rows = sweep_group_rows_for(h, Hp, bwd);
        s.dh_ext = bwd ? md.dh_ext : nullptr;
        s.counters = h->sweep_cnt; s.status = h->sweep_status; s.spin_ticks = h->spin_ticks; s.poll_mask = h->poll_mask;
        s.flag_stride = h->flag_stride;
        s.xcc_tab = h->sweep_cnt + (size_t)((Bp + 7) / 8) * h->T * h->flag_stride; s.xcd_fast = bwd ? (h->xcd_fast >> 1) & 1 : h->xcd_fast & 1;
        s.xchg = h->sweep_xchg;
        s.n_valid = (h->dt == F32 && h->f32_valu) ? h->rows_in_use : 0;   // the kernel the planning path runs at this batch
        s.stash_via_lds = bwd ? (h->dt == F32 ? (h->wide_ingest ? 2 : 0) : (h->own_store ? 2 : 0)) : (h->stash_lds ? 1 : 0);
        // ... with the arguments the planning path gives it: the predictor's backward sweep carries the ride-along tile (and, by default, leaves dA unwritten)
        if (bwd && &md == &h->pred && h->bwd_xt != 0 && ly.WihT && lstm_rs_ride_along_supported(Hp, ly.in_p) && Tl == h->T && pred_ride_along(h)) set_ride_along(h, ly, s);
        // the sweep alone between the events of every repetition (the zeroing of its flag slice, a kernel of its own outside an iteration, stays outside)
        std::vector<hipEvent_t> ev(2 * (size_t)reps);
        for (auto& e : ev) PL_HIP(hipEventCreate(&e));
        for (int i = 0; i < reps; ++i) {
            s.counters = take_sweep_slice(h, h->stream);
            s.xcc_tab = s.counters + (size_t)((Bp + 7) / 8) * h->T * h->flag_stride;
            PL_HIP(hipEventRecord(ev[2 * i], h->stream));
            launch_sweep(h, h->stream, bwd, Hp, grid, s);
            PL_HIP(hipEventRecord(ev[2 * i + 1], h->stream));
        }
        PL_HIP(hipEventSynchronize(ev.back()));
        float ms = 0.f;
        for (int i = 0; i < reps; ++i) {
            float m1 = 0.f;
            PL_HIP(hipEventElapsedTime(&m1, ev[2 * i], ev[2 * i + 1]));
            ms += m1;
        }
        for (auto& e : ev) (void)hipEventDestroy(e);
        *avg_ms_out = ms / reps;
        if (flops_per_launch_out) *flops_per_launch_out = 2.0 * h->B * 4.0 * md.H * md.H * (Tl - 1);
        return check_launch();
    }
#ifdef PL_STAMPS
    // diagnostic build: every launch writes [block][8] s_memrealtime stamps; dumped raw to $PL_STAMP_FILE
    const int nblk = kernel == PL_KERNEL_LSTM_FWD_STEP ? (Hp / 16) * ((Bp + 63) / 64) : (Hp / 32) * ((Bp + 31) / 32);
    unsigned long long* stamps = nullptr;
    PL_HIP(hipMalloc(&stamps, sizeof(unsigned long long) * 8 * nblk * reps));
    PL_HIP(hipMemset(stamps, 0, sizeof(unsigned long long) * 8 * nblk * reps));
#endif
    hipEvent_t e0, e1;
    PL_HIP(hipEventCreate(&e0));
    PL_HIP(hipEventCreate(&e1));
    PL_HIP(hipEventRecord(e0, h->stream));
    for (int i = 0; i < reps; ++i) {
        const int t = 1 + i % (Tl - 2);   // interior steps: every operand present
        LstmStepArgs s{};
        s.Bp = Bp;
        s.Hp = Hp;
        s.G_t = off(ly.G, (size_t)t * Bp * 4 * Hp, a);
        s.c_stash_t = off(ly.c, (size_t)t * Bp * Hp, a);
#ifdef PL_STAMPS
        s.stamps = stamps + (size_t)i * nblk * 8;
#endif
        if (kernel == PL_KERNEL_LSTM_FWD_STEP) {
            s.W = ly.Whh;
            s.h_prev = off(ly.h, (size_t)(t - 1) * Bp * Hp, a);
            s.h_out = off(ly.h, (size_t)t * Bp * Hp, a);
            s.c_in = h->c_run[(t - 1) & 1];
            s.c_out = h->c_run[t & 1];
            launch_lstm_fwd_step(h->stream, h->dt, s);
        } else {
            s.W = ly.WhhT;
            s.G_next = off(ly.G, (size_t)(t + 1) * Bp * 4 * Hp, a);
            s.c_in = h->dc_run[(t + 1) & 1];
            s.c_out = h->dc_run[t & 1];
            s.c_stash_prev = off(ly.c, (size_t)(t - 1) * Bp * Hp, a);
            s.dh_ext = off(md.dh_ext, (size_t)t * Bp * Hp, a);
            launch_lstm_bwd_step(h->stream, h->dt, s);
        }
    }
    PL_HIP(hipEventRecord(e1, h->stream));
    PL_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    PL_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
#ifdef PL_STAMPS
    if (const char* path = std::getenv("PL_STAMP_FILE")) {
        std::vector<unsigned long long> host((size_t)8 * nblk * reps);
        PL_HIP(hipMemcpy(host.data(), stamps, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::string fn = std::string(path) + (kernel == PL_KERNEL_LSTM_FWD_STEP ? ".fwd" : ".bwd");
        if (FILE* f = std::fopen(fn.c_str(), "wb")) {
            const int hdr[2] = {nblk, reps};
            std::fwrite(hdr, sizeof(int), 2, f);
            std::fwrite(host.data(), sizeof(unsigned long long), host.size(), f);
            std::fclose(f);
        }
    }
    (void)hipFree(stamps);
#endif
    *avg_ms_out = ms / reps;
    if (flops_per_launch_out) *flops_per_launch_out = 2.0 * h->B * 4.0 * md.H * md.H;
    return check_launch();
}

int64_t pl_device_bytes(const pl_handle* h) { return h ? (int64_t)h->bytes : 0; }

// prefetcher workgroups beside the predictor's whole-sequence backward sweep (0: another kernel carries it, or none fit)
static int bwd_prefetchers_planned(const pl_handle* h) {
    pl_handle* hm = const_cast<pl_handle*>(h);
    if (h->dt != BF16 || !h->use_sweep || h->fused_bwd_ok || h->bwd_mode != 1 || !h->sweep_xchg || h->bwd_stream != 1 || h->bwd_waves == 4 ||
        h->bwd_dma != 0 || h->bwd_chains > 0 || use_sweep16(hm, h->pred.Hp, true))
        return 0;
    if (h->pred.L > 1 && (h->wf_dirs & 2) && wavefront_chunks(hm, h->pred, h->pred.Tl, 0) > 0) return 0;   // time chunks keep the whole-workgroup hand-off: no prefetchers
    const int grid = sweep_grid_for(hm, h->pred.Hp, true);
    return grid > 0 ? bwd_prefetchers_for(h, h->pred.Hp, grid) : 0;
}

int pl_plan_info(const pl_handle* h, int32_t* out, int n) {
    if (!h || !out || n < 0) return fail(PL_ERR_INVALID, "pl_plan_info: null handle / output");
    const int32_t v[PL_PLAN_COUNT] = {
        h->fused_fwd_ok ? 1 : 0,
        h->fused_bwd_ok ? 1 : 0,
        h->fused_fwd_ok ? h->fused_Cp : 0,
        h->fused_fwd_ok ? h->fused_Ce : 0,
        h->fused_bwd_ok ? h->fused_Cp_bwd : 0,
        h->fused_bwd_ok ? h->fused_Ce_bwd : 0,
        h->fused_fwd_ok ? h->fused_active_fwd : 0,
        h->fused_bwd_ok ? h->fused_active_bwd : 0,
        h->bwd_waves,
        h->n_cu,
        g_retained_branched_execs.load(std::memory_order_relaxed),
        h->fused_rows16 ? 16 : ((h->fused_fwd_ok || h->fused_bwd_ok) ? 32 : 0),
        h->fused_fwd_ok ? (h->fused_fwd2 ? 2 : 1) : 0,
        bwd_prefetchers_planned(h),
    };
    for (int i = 0; i < n && i < PL_PLAN_COUNT; ++i) out[i] = v[i];
    for (int i = PL_PLAN_COUNT; i < n; ++i) out[i] = 0;
    return PL_OK;
}

double pl_flops_per_iteration(const pl_handle* h) {
    if (!h) return 0.0;
    // forward + backward-data = 2 x forward; elementwise work excluded
    double f = (lstm_flops_per_step(h->pred.L, h->pred.H, h->C) + 2.0 * h->pred.H * h->M) * h->T;
    if (h->need_emb_in_step())
        f += lstm_flops_per_step(h->emb.L, h->emb.H, h->M) * h->Tp +
             (h->emb_post > 0 ? 2.0 * h->emb_post * (h->emb.H + h->S) : 2.0 * h->emb.H * h->S);
    if (h->tube_on())
        f += (lstm_flops_per_step(h->tube.L, h->tube.H, h->C) + 2.0 * h->tube.H * h->U) * h->T +
             (lstm_flops_per_step(h->tmel.L, h->tmel.H, h->U) + 2.0 * h->tmel.H * h->M) * h->T +
             lstm_flops_per_step(h->temb.L, h->temb.H, h->U) * h->T + 2.0 * h->temb.H * h->S;
    return 2.0 * f * h->B;
}

int pl_debug_read(pl_handle* h, const char* name, float* out, int64_t max_elems, int64_t* n_out) {
    if (!h || !name) return fail(PL_ERR_INVALID, "pl_debug_read: NULL argument");
    DeviceGuard guard(h->cfg.device);
    const std::string nm(name);
    const void* src = nullptr;
    int64_t n = 0;
    int kind = 0;   // 0 activation type, 1 f32, 2 f64
    auto model_buf = [&](Model& md, const std::string& rest) -> bool {
        if (rest == "dh_ext") { src = md.dh_ext; n = (int64_t)md.Tl * h->Bp * md.Hp; kind = 0; return true; }
        if (rest == "gWlin" && md.train_ready) { src = md.gWlin; n = (int64_t)md.out_p * md.Hp; kind = 1; return true; }
        if (rest == "gblin" && md.train_ready) { src = md.gblin; n = (int64_t)md.out_p; kind = 1; return true; }
        if (rest.size() < 2) return false;
        const int l = std::atoi(rest.c_str() + 1);
        if (l < 0 || l >= md.L) return false;
        LstmLayer& ly = md.layers[l];
        switch (rest[0]) {
            case 'G': src = ly.G; n = (int64_t)md.Tl * h->Bp * 4 * md.Hp; kind = 0; return true;
            case 'h': src = ly.h; n = (int64_t)md.Tl * h->Bp * md.Hp; kind = 0; return true;
            case 'c': src = ly.c; n = (int64_t)md.Tl * h->Bp * md.Hp; kind = 0; return true;
            case 'W': src = ly.Whh; n = (int64_t)4 * md.Hp * md.Hp; kind = 0; return true;
            case 'b': src = ly.bias; n = (int64_t)4 * md.Hp; kind = 1; return true;
            // weight gradients of the last pl_train_pred_step, padded compute layout [4Hp][in_p] / [4Hp][Hp] / [4Hp]
            case 'i': if (!md.train_ready) return false; src = ly.gWih; n = (int64_t)4 * md.Hp * ly.in_p; kind = 1; return true;
            case 'r': if (!md.train_ready) return false; src = ly.gWhh; n = (int64_t)4 * md.Hp * md.Hp; kind = 1; return true;
            case 'd': if (!md.train_ready) return false; src = ly.gb; n = (int64_t)4 * md.Hp; kind = 1; return true;
            default: return false;
        }
    };
    bool ok = false;
    if (nm == "pred.G0" && h->pred_dA_skipped)
        return fail(PL_ERR_STATE, "pl_debug_read: the last planning iteration did not keep the predictor's dA (dL/dCP rode along in the sweep); PAULE_HIP_BWD_XT=1 keeps it");
    if (nm.rfind("pred.", 0) == 0) ok = model_buf(h->pred, nm.substr(5));
    else if (nm.rfind("emb.", 0) == 0 && h->emb.L > 0) ok = model_buf(h->emb, nm.substr(4));
    else if (nm == "X0") { src = h->X0; n = (int64_t)h->T * h->Bp * h->Cp; kind = 0; ok = true; }
    else if (nm == "Y") { src = h->Y; n = (int64_t)h->T * h->Bp * h->Mp; kind = 1; ok = true; }
    else if (nm == "mel") { src = h->mel_bm; n = (int64_t)h->B * h->Tp * h->M; kind = 1; ok = true; }
    else if (nm == "mel_tm") { src = h->mel_tm; n = (int64_t)h->Tp * h->Bp * h->Mp; kind = 0; ok = true; }
    else if (nm == "dY") { src = h->dY; n = (int64_t)h->T * h->Bp * h->Mp; kind = 0; ok = true; }
    else if (nm == "dX") { src = h->dX; n = (int64_t)h->T * h->Bp * h->Cp; kind = 1; ok = true; }
    else if (nm == "dX2" && h->dX2) { src = h->dX2; n = (int64_t)h->T * h->Bp * h->Cp; kind = 1; ok = true; }
    else if (nm == "sem" && h->sem) { src = h->sem; n = (int64_t)h->Bp * h->Sp; kind = 1; ok = true; }
    else if (nm == "dsem" && h->dsem) { src = h->dsem; n = (int64_t)h->Bp * h->Sp; kind = 0; ok = true; }
    else if (nm == "dv" && h->dv) { src = h->dv; n = (int64_t)h->Bp * h->emb.Hp; kind = 0; ok = true; }
    else if (nm == "dmel_e" && h->dmel_e) { src = h->dmel_e; n = (int64_t)h->Tp * h->Bp * h->Mp; kind = 1; ok = true; }
    else if (nm == "grad") { src = h->grad; n = (int64_t)h->B * h->T * h->C; kind = 2; ok = true; }
    else if (nm == "x") { src = h->x; n = (int64_t)h->B * h->T * h->C; kind = 2; ok = true; }
    else if (nm == "m") { src = h->m; n = (int64_t)h->B * h->T * h->C; kind = 2; ok = true; }
    else if (nm == "v") { src = h->v; n = (int64_t)h->B * h->T * h->C; kind = 2; ok = true; }
    else if (nm == "scal") { src = h->scal; n = (int64_t)h->B * 8; kind = 2; ok = true; }
    if (!ok) return fail(PL_ERR_INVALID, "pl_debug_read: unknown buffer '" + nm + "'");
    if (n_out) *n_out = n;
    if (!out) return PL_OK;
    if (max_elems < n) return fail(PL_ERR_INVALID, "pl_debug_read: output too small");
    if (kind == 0) launch_act_to_f32(h->stream, h->dt, src, out, n);
    else if (kind == 1) PL_HIP(hipMemcpyAsync(out, src, sizeof(float) * n, hipMemcpyDeviceToDevice, h->stream));
    else launch_f64_to_f32(h->stream, static_cast<const double*>(src), out, n);
    return check_launch();
}

}  // extern "C"
