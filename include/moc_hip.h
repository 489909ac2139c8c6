/*
 * moc_hip.h -- C ABI of libmoc_hip.so: the MI355X (gfx950) kernels for the hot
 * path of xmed-lab/MOC.
 *
 * The reference has no FFI of its own: its seam is a set
 * of Python callables in main_moc.py and utils/patch_selection_classifier*.py.
 * Each entry point below names the reference lines it replaces; the Python
 * package moc_amd binds them with ctypes (INTEGRATION.md shows the stub a
 * maintainer of the reference would add).
 *
 * Conventions
 *   - every pointer marked "device" is HIP device memory owned by the caller
 *     (PyTorch tensors in practice); the compute entry points never allocate, free or
 *     keep device memory and hold no state between calls.  The ONE exception is the
 *     node-local exchange handle `moc_p2p_t` (moc_p2p_create ... moc_p2p_destroy, below):
 *     it owns its fine-grained receive buffers (hipExtMallocWithFlags), one pinned host
 *     error word, the peers' IPC mappings and a sequence counter -- allocation and state
 *     are confined to that handle, created and destroyed explicitly by the caller, and
 *     exist only in a multi-rank run ("no cross-call state except an
 *     optional handle for persistent-kernel resources");
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *     every call returns without synchronising the host;
 *   - return value 0 = success; otherwise a MOC_E* code, with a message
 *     available from moc_last_error() (thread local);
 *   - all matrices are row-major and dense; bag rows must be 16-byte aligned
 *     (D*elem_size % 16 == 0 and a 16-byte aligned base);
 *   - "slot space": work arrays are indexed by slot, not by row of X; total_rows is the
 *     number of slots (sum of the batch's slide sizes).  The kept (un-masked) rows of slide b occupy slots
 *     [row_off[b], row_off[b] + n_kept[b]) of every per-row work array, in
 *     ascending row order; with no mask n_kept[b] == rows of slide b.
 */
#ifndef MOC_HIP_H
#define MOC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOC_ABI_VERSION 18

enum { MOC_TICKET_QUEUES = 64, MOC_TICKET_STRIDE = 64,     /* moc_batch_t.tile_ticket: counters, int32 words between them */
       MOC_TICKET_WORDS = (64 + 8) * 64 };

enum { MOC_OK = 0, MOC_EINVAL = 1, MOC_EUNSUPPORTED = 2, MOC_ELAUNCH = 3 };

/* storage type of the bag X; arithmetic is always fp32-accumulate */
enum { MOC_F32 = 0, MOC_BF16 = 1, MOC_F16 = 2 };

/* moc_batch_t.flags.  MOC_STATS_COMPACT: the score pass writes C+5 statistics per row instead of 2C+3 -- the C softmax
 * columns (patch_selection_classifier_index.py:34) are not stored; their consumers (moc_select's psi_sigma keys,
 * moc_gather_candidates' s_sigma) re-form softmax[c] = exp2((logit[c] - m1) * log2 e) * (1/den) from the row maximum m1
 * and the reciprocal of the denominator, with the score pass's own arithmetic: the same bits.  For wide banks, where the
 * statistics are a quarter of the pass's HBM traffic (thirty classes: 253 -> 141 bytes per 1 KiB row).  moc_row_stats
 * and callers that read `stats` themselves (the zero-shot poolings) use the full layout. */
enum { MOC_STATS_COMPACT = 1,
       /* moc_select: one workgroup per column even for wide banks (the default there is one workgroup per slide and
        * group of eight columns); both give the same flags -- this bit exists so that tests can say so */
       MOC_SELECT_PER_COLUMN = 2,
       /* evaluation passes: moc_gather_candidates does not materialise the [2C+2, S] candidate columns of wide banks
        * (C > 4) and moc_meta_forward reads a selected row's four scores (main_moc.py:359-366, :482-492) straight from
        * `stats` through sel_idx -- the same values.  Entry points that need `cand` itself (the train steps,
        * moc_mix_fixed, moc_pack_selected*) refuse such a batch. */
       MOC_CAND_FROM_STATS = 4,
       /* moc_meta_forward over many slides: keep to 64 rows per workgroup (the default for launches of at least four
        * slides with 1024 or more selectable rows on 16-bit bags is 128 rows per workgroup, rows by LDS-DMA); both give
        * the bits of the one-slide kernel -- the bit exists so that tests can say so */
       MOC_FORWARD_ROWS64 = 8,
       /* moc_meta_forward on fp32 bags, 16 rows per workgroup: keep to four waves, every wave the whole chain over the
        * columns of its hidden units (the default splits the columns over four wave groups, sixteen waves: the training
        * step's forward); the same bits -- the bit exists so that tests can say so */
       MOC_FORWARD_FOUR_WAVES = 16,
       /* moc_meta_forward of ONE slide on 16-bit bags: keep to 16 rows per workgroup even when the slide has 16,384 or
        * more selectable rows (the default there is the 64-row kernel: every W1 fragment feeds four row tiles -- at
        * 25,000 rows the sixteen-row workgroups re-read 390 KB of W1 image 1,580 times); the same bits -- the bit exists
        * so that tests can say so */
       MOC_FORWARD_ROWS16 = 32 };

/* bits of `discard_bits`, in the order of main_moc.py:341-350 */
enum { MOC_SEL_TOPK = 1, MOC_SEL_DELTA_SOFTMAX = 2, MOC_SEL_DELTA_DIFF = 4, MOC_SEL_BOTTOMK = 8 };

typedef void* moc_stream_t; /* hipStream_t */

/* A batch of slides packed back to back in one [total_rows, D] array, plus the
 * caller-allocated work arrays every stage reads/writes.  `stride` below is
 * total_rows. */
typedef struct moc_batch {
    /* ---- inputs ---- */
    const void*    X;          /* device [total_rows, D]                                  */
    int32_t        dtype;      /* MOC_F32 | MOC_BF16 | MOC_F16 (storage; arithmetic is fp32) */
    int32_t        D;          /* embedding dim (512 for CONCH)                          */
    int64_t        total_rows;
    int32_t        n_slides;
    int32_t        max_rows;   /* host bound on the rows a slide brings to the selectors (sizes grids and picks kernel
                                  shapes): max rows of any slide, or -- tighter -- max KEPT rows (moc_host_max_kept) */
    const int64_t* row_off;    /* device [n_slides+1], first SLOT of each slide (prefix
                                  sum of the slide sizes)                                */
    const int64_t* row_off_host; /* host copy of row_off, or NULL.  With it, single-slide launches
                                  (the sequential train steps) get the slide's first slot as a
                                  kernel argument instead of a dependent device load          */
    const int64_t* x_off;      /* device [n_slides], first row of each slide inside X, or
                                  NULL = row_off (slides packed in batch order).  Lets a
                                  batch visit resident slides in any order / repeatedly
                                  (dataset_generic.py:380-393 repeat_num) without copies */
    const uint8_t* mask;       /* [total_rows] 0/1 keep flags in device OR device-mapped pinned host memory
                                  (read once, by moc_mask_compact), or NULL = keep all
                                  (main_moc.py:329-331; drawn by the host, see moc_amd)  */
    int32_t        C;          /* n_classes                                              */
    int32_t        Ce;         /* columns of zeroshot_weights_ext (C + background)       */
    int32_t        topj;       /* --topj                                                 */
    int32_t        topk;       /* --topk                                                 */
    uint32_t       discard_bits;
    uint32_t       flags;      /* MOC_STATS_COMPACT or 0 (the field was `reserved`, always 0, before round 3) */
    /* ---- work arrays (device) ---- */
    int32_t* kept;       /* [total_rows + 16] slot -> row index inside its slide (unused if mask==NULL);
                            the 16 entries of slack let the score pass read a tile's 16 indices
                            with one scalar load without running off the allocation            */
    int32_t* n_kept;     /* [n_slides]                                                               */
    float*   stats;      /* [2C+3, total_rows] per-slot: logits[C] | softmax[C] | gap | bg_sum | bg_max;
                            with MOC_STATS_COMPACT only the first C+5 rows are used:
                            logits[C] | m1 | 1/den | gap | bg_sum | bg_max                                 */
    uint8_t* sel_flag;   /* [total_rows]      union membership                                        */
    int32_t* sel_idx;    /* [total_rows]      selected_index of slide b at [row_off[b], +n_sel[b])     */
    int64_t* sel_row;    /* [total_rows]      same positions: row of X (packed) to gather             */
    int32_t* n_sel;      /* [n_slides]        S                                                       */
    float*   cand;       /* [2C+2, total_rows] per selected row: s_p[C] | s_sigma[C] | s_delta | s_beta */
    /* ---- placement of the score pass (round 3; both nullable = static walk over the whole chip) ---- */
    const uint32_t* cu_reserved; /* device [128]: compute units the score pass stays off (see moc_cu_census)   */
    int32_t*        tile_ticket; /* device [MOC_TICKET_WORDS]: the score pass's tile counters                  */
    /* ---- round 4, second session (nullable) ---- */
    const int32_t*  n_sel_host;  /* HOST [n_slides]: a copy of n_sel that the caller has seen arrive (e.g. an asynchronous
                                    copy behind phase A whose event has completed), or NULL.  The train steps then take
                                    a slide's S as a kernel argument instead of loading it -- one dependent round trip
                                    less at the head of the forward (arguments -> sel_row -> rows), exact grids, fewer
                                    record keys per lane in the step.  The same results either way.                   */
} moc_batch_t;

/* The meta-learner ("senet", main_moc.py:299-312) and its Adam state
 * (main_moc.py:316; torch.optim.Adam defaults otherwise).  All device fp32. */
typedef struct moc_meta {
    float *W1, *b1, *W2, *b2;         /* [H,D] [H] [4,H] [4]   H = 64                      */
    float *m_W1, *m_b1, *m_W2, *m_b2; /* exp_avg      (may be NULL for forward-only use)   */
    float *v_W1, *v_b1, *v_W2, *v_b2; /* exp_avg_sq                                        */
    float *g_W1, *g_b1, *g_W2, *g_b2; /* gradient outputs (moc_train_grad), may be NULL    */
    void  *W1_image;                  /* device scratch, moc_w1_image_bytes(D, dtype): W1 re-laid in
                                         MFMA operand order for the forward pass.  Derived data the
                                         library rebuilds / keeps in sync itself; contents need not
                                         survive between calls.                                */
    double lr, beta1, beta2, eps, weight_decay; /* the optimizer's Python floats, unrounded     */
    int32_t H;                        /* hidden width, must be 64                          */
    int32_t D;                        /* input width (== batch D)                          */
    int64_t step;                     /* Adam steps already taken                          */
} moc_meta_t;

/* Work arrays of the meta-learner stage, caller allocated (device). */
typedef struct moc_meta_ws {
    float*   H1;        /* [total_rows, H]  relu(W1 x + b1) of every selected row            */
    float*   gates;     /* [total_rows, 4]  lambda                                           */
    float*   mixed;     /* [C, total_rows]  gated sum of the candidates ("final_logits")     */
    float*   pooled;    /* [n_slides, C]    top-K mean per class                             */
    int32_t* topk_idx;  /* [n_slides, C, topk] positions (0..S) of the pooled rows, by value */
    int32_t* topk_cnt;  /* [n_slides, C]    min(K, S)                                        */
    float*   loss;      /* [n_slides]       cross entropy                                    */
    int32_t* pred;      /* [n_slides]       argmax of pooled                                 */
    float*   pair_dh;   /* [C*topk, H]      backward scratch (one slide at a time)           */
    float*   W2_alt;    /* [4, H]           second copy of W2: the one-launch step reads one and writes
                                            the other (may be NULL: three-launch step is used)      */
    int64_t* pair_row;  /* [C*topk]                                                          */
    int32_t* n_pair;    /* [1]                                                               */
    /* round 4: tile records (nullable).  Scratch of moc_tile_ws_bytes(total_rows, n_slides, C) bytes, 16-byte aligned:
     * the forward of a training step leaves, per 16-row tile and class, the tile's four largest mixed scores with
     * their rows' ids, gates and candidate scores; the one-launch step then pools among those records instead of
     * re-reading and ranking every mixed score (exactness checked in the kernel, full scores as the fall-back).
     * NULL: the round-3 step. */
    void*    tile_ws;
    int64_t  tile_ws_bytes;
} moc_meta_ws_t;

int         moc_version(void);
const char* moc_last_error(void);
size_t      moc_w1_image_bytes(int D, int dtype);
size_t      moc_tile_ws_bytes(int64_t total_rows, int n_slides, int C);   /* moc_meta_ws_t.tile_ws */

/* ---- classifier bank ------------------------------------------------------
 * Re-lays [zeroshot_weights | zeroshot_weights_ext[:, C:]] (main_moc.py:336-337:
 * the foreground columns come from W, the background ones from W_ext) into the
 * MFMA operand image the score kernel streams from LDS.  For MOC_BF16 bags the
 * fp32 weights are split into three bf16 terms so products stay exact in fp32.
 * Redo whenever the weights change.  `fg_from_ext` != 0 takes the foreground
 * columns from W_ext[:, :C] instead (bottomk_irrel_classifier_pooling,
 * utils/patch_selection_classifier.py:151). */
size_t moc_bank_bytes(int D, int Ce, int dtype);
int    moc_prepare_bank(const float* W /*device [D,C]*/, const float* W_ext /*device [D,Ce]*/,
                        int D, int C, int Ce, int dtype, int fg_from_ext,
                        void* bank_out /*device, moc_bank_bytes*/, moc_stream_t stream);

/* ---- phase A: parameter-free part of slide_process (main_moc.py:322-366) ---- */

/* a1  row mask -> kept list + n_kept (main_moc.py:329-331).  No-op when mask==NULL. */
int moc_mask_compact(const moc_batch_t* B, moc_stream_t stream);

/* a1, host side: the row masks of `n` consecutive rows exactly as main_moc.py:330 draws them
 * (`torch.rand(N) > 0.5`, CPU default generator = mt19937, one output per sample).  rng_state is
 * the byte image of torch.get_rng_state() (in/out: advanced by n draws, hand it back with
 * torch.set_rng_state); out gets n 0/1 bytes.  Returns the number of kept rows, -1 on error.
 * Pure host code: same bits as torch.rand, no GPU involved. */
int64_t moc_host_draw_masks(uint8_t* rng_state, int64_t state_bytes, int64_t n, uint8_t* out);

/* Largest number of non-zero flags of any slide, `mask` being the host keep flags of a batch laid out by
 * row_off_host[n_slides + 1]: a tight `max_rows` for a masked batch (the selectors only ever see the kept rows of
 * `feat[mask]`, main_moc.py:329-331), which lets wide shapes pool inside the step kernel.  Host only; -1 on bad
 * arguments. */
int64_t moc_host_max_kept(const uint8_t* mask, const int64_t* row_off_host, int n_slides);

/* a2 + per-row parts of a4-a6, a9: X.[W|W_ext] and the row statistics
 * (main_moc.py:336-337, :360-365; patch_selection_classifier_index.py:34, :46-48, :75).
 * Also clears sel_flag. */
int moc_scores(const moc_batch_t* B, const void* bank, moc_stream_t stream);
/* moc_scores with two hipEvent_t (created with timing enabled) that receive the score kernel's own start and end
 * time stamps (hipExtLaunchKernel): hipEventElapsedTime(start, stop) is then the kernel's duration as a profiler
 * reports it, whatever else the GPU is running.  bench.py's `roofline` uses it. */
int moc_scores_timed(const moc_batch_t* B, const void* bank, moc_stream_t stream, void* start_event, void* stop_event);

/* a2 from a cache (round 4, opt-in): the statistics of a row depend only on the row and the frozen bank, so a resident
 * split can keep the statistics of ALL its rows (`stats_all` [rows of stats, all_rows], laid out like `stats` and indexed
 * by the row's position in B->X: the output of an UNMASKED moc_scores over the split with the same B->flags layout) and a
 * pass then copies the kept rows' statistics into slot order instead of reading the bags again (main_moc.py:336-337 is
 * recomputed per visit in the reference; the bits are the same).  Needs moc_mask_compact's lists for a masked batch. */
int moc_scores_from_cache(const moc_batch_t* B, const float* stats_all, int64_t all_rows, moc_stream_t stream);

/* The same statistics from an already computed logits matrix (device [N, Ct] row-major,
 * first C columns foreground): stats_out is [2C+3, N].  For callers that hold logits, not
 * bags (index_*_classifier / *_pooling, utils/patch_selection_classifier*.py).  With C == 1
 * the gap column is +inf (the reference's topk(2) would raise). */
int moc_row_stats(const float* logits, int64_t N, int Ct, int C, float* stats_out, moc_stream_t stream);

/* a3-a7: the four top-j selectors and their union, as flags
 * (patch_selection_classifier_index.py:17-87, main_moc.py:341-352). */
int moc_select(const moc_batch_t* B, moc_stream_t stream);

/* a7-a9: ascending compaction of the union -> sel_idx/sel_row/n_sel, candidate
 * scores of the selected rows, optionally the gathered rows themselves
 * (main_moc.py:354-366).  selected_feat: device [total_rows, D] in X's dtype or NULL. */
int moc_gather_candidates(const moc_batch_t* B, void* selected_feat, moc_stream_t stream);

/* all four above, in order */
int moc_phase_a(const moc_batch_t* B, const void* bank, moc_stream_t stream);

/* Phase A's result for slides [slide0, slide0+n) in a fixed-capacity, position-independent form -- what
 * slide_process hands to the meta-learner (main_moc.py:367-375: selected_feat + the four candidate matrices),
 * `cap` rows per slide: feat_out device [n][cap][D] in X's dtype, cand_out device [n][2C+2][cap] fp32; entries at
 * and beyond n_sel[slide] are not written.  The exact-sequential multi-GPU mode (SURVEY.md section 8e mode 1)
 * all-gathers these blocks so that every GPU runs the reference's one-Adam-step-per-slide recurrence
 * (main_moc.py:380-410) over slides whose phase A ran elsewhere. */
int moc_pack_selected(const moc_batch_t* B, int slide0, int n, int cap, void* feat_out, float* cand_out,
                      moc_stream_t stream);

/* The same hand-over without padding, for a gather of unequal pieces: slide slide0+b's min(n_sel, cap) rows start at
 * row sum_{q<b} min(n_sel[slide0+q], cap) of feat_out (device [out_rows][D], X's dtype), and its candidate scores are
 * the rows cand_out[that row + t][2C+2] (device fp32, ROW-major: s_p[C] | s_sigma[C] | s_delta | s_beta per selected
 * row -- main_moc.py:359-366's four matrices side by side); rows at and beyond out_rows are dropped.  What the
 * exact-sequential mode sends since round 3: sum S rows per rank instead of n x cap (44 % less at NSCLC-16). */
int moc_pack_selected_rows(const moc_batch_t* B, int slide0, int n, int cap, void* feat_out, float* cand_out,
                           int64_t out_rows, moc_stream_t stream);

/* ---- phase B: meta-learner, pooling, loss, update ------------------------- */

/* use_bits: which of the four gated terms enter the sum (bit i = term i).
 * train: ~discard_bits & 15 (main_moc.py:396-403); eval: see moc_amd.main_moc
 * for the reference's quirk (main_moc.py:486-492). */

/* a10-a11 for slides [slide0, slide0+n): H1, gates, mixed.  ws->H1 / ws->gates may be NULL (evaluation:
 * only the backward pass reads them; 256 + 16 bytes per selected row not written). */
int moc_meta_forward(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws,
                     int slide0, int n, uint32_t use_bits, moc_stream_t stream);

/* ablation_evaluation's parameter-free mixes (main_moc.py:538-553) in place of
 * moc_meta_forward: mode 0 = avg (0.25 each), 1 = sum, 2 = max of the four candidates. */
int moc_mix_fixed(const moc_batch_t* B, const moc_meta_ws_t* ws, int slide0, int n, int mode,
                  moc_stream_t stream);

/* a12-a13 (+a16 argmax) for slides [slide0, slide0+n): pooled, topk_idx, loss, pred.
 * labels: device int64 [n_slides]. */
int moc_pool_loss(const moc_batch_t* B, const moc_meta_ws_t* ws, const int64_t* labels,
                  int slide0, int n, moc_stream_t stream);

/* a13 + a16 alone: cross entropy and argmax of n rows of pooled logits [n, C]
 * (F.cross_entropy(logits, lbl) and logits.argmax(dim=1), main_moc.py:433-434, :494-495). */
int moc_ce_loss(const float* pooled, const int64_t* labels, int n, int C, float* loss, int32_t* pred,
                moc_stream_t stream);

/* Batched runs (round 4): `n_runs` independent meta-learners stepped in lockstep by ONE forward launch and ONE step launch
 * per meta-step (grid.y / grid.z = run) -- the folds x shots of the reference's launcher (scripts/moc_train.sh:11-31) as
 * one process per GPU instead of one process per run.  Run r owns the slides slide0 + r * slide_stride ... of B, its tensors
 * lie par_stride floats behind run 0's in ONE arena per kind (W1 | b1 | W2 | b2 of a run inside one block, the moments
 * laid out alike), its W1 operand image image_stride bytes behind run 0's.  All runs share the Adam hyper-parameters and
 * step count of M.  Per run the arithmetic is moc_train_steps's: bit-identical parameters. */
typedef struct moc_runs {
    int32_t n_runs;        /* 1 .. 16 */
    int32_t slide_stride;  /* slides of B between consecutive runs' first slides (>= n) */
    int64_t par_stride;    /* floats */
    int64_t image_stride;  /* bytes, >= moc_w1_image_bytes(D, dtype) */
} moc_runs_t;
/* M: run 0's tensors.  ws->W2_alt: [n_runs, 4, H].  With ws->tile_ws and a shape of the tile-record step (C <= 16, K <= 16,
 * C K <= 64, <= 4,096 selectable rows) the runs step in lockstep (moc_train_runs_mode == 1); other shapes of the one-launch
 * steps (wide banks: EBRAINS-30, the 64-way shape) take every run's pass one after the other on `stream`, with the launches
 * of moc_train_steps (mode 2: give every run a call and a stream of its own to have their chains side by side); shapes of
 * the three-launch step are refused (mode 0: its scratch is one per batch). */
int moc_train_steps_runs(const moc_batch_t* B, const moc_meta_t* M, const moc_runs_t* R, const moc_meta_ws_t* ws,
                         const int64_t* labels, int slide0, int n, uint32_t use_bits, moc_stream_t stream);
int moc_train_runs_mode(const moc_batch_t* B, const moc_meta_ws_t* ws);

/* a10-a14 for ONE slide without the update: forward, pooling, loss (ws->loss/pooled/pred) and
 * the gradients of that loss w.r.t. the four parameter tensors, written (not accumulated) to
 * M->g_*.  The data-parallel step all-reduces them and calls moc_adam_step. */
int moc_train_grad(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws,
                   const int64_t* labels, int slide, uint32_t use_bits, moc_stream_t stream);

/* a14 for a caller-written loop (main_moc.py:390-410 kept as it is around slide_process): the backward of
 * `lambdas = model(selected_feat)` given d loss / d lambdas.  X [S, D] are the rows the forward ran on (storage `dtype`),
 * H1 [S, H] / gates [S, 4] what moc_meta_forward left for them, grad_gates [S, 4] autograd's gradient (rows of zeros
 * carry nothing and are skipped).  Writes (does not accumulate) g_W1 [H, D], g_b1 [H], g_W2 [4, H], g_b2 [4].
 * Scratch, all device: pair_dz [S, 4], pair_dh [S, H], pair_row [S], n_pair [1]. */
int moc_senet_backward(const void* X, int dtype, int64_t S, int D, const float* H1, const float* gates,
                       const float* grad_gates, const float* W2, float* g_W1, float* g_b1, float* g_W2, float* g_b2,
                       float* pair_dz, float* pair_dh, int64_t* pair_row, int32_t* n_pair, moc_stream_t stream);

/* a15: one Adam step (coupled L2) from M->g_* scaled by grad_scale; uses step = M->step+1. */
int moc_adam_step(const moc_meta_t* M, float grad_scale, moc_stream_t stream);

/* e (data-parallel training): `n` synchronous steps on THIS rank's slides slide0..slide0+n-1.
 * Per step: forward, pooling, loss and gradients (as moc_train_grad), ONE all-reduce of the
 * flat gradient, the Adam step with grad_scale = 1/world (as moc_adam_step, step = M->step+1+t).
 * The collective is the caller's: `allreduce` has ncclAllReduce's signature (RCCL is not linked
 * into this library) and is called as allreduce(grad_flat, grad_flat, grad_count, ncclFloat32,
 * ncclSum, comm, stream); M->g_* must point into grad_flat.  NULL allreduce with world == 1
 * runs the same step sequence without a collective.  M->step itself is not advanced. */
typedef int (*moc_allreduce_fn)(const void* sendbuf, void* recvbuf, size_t count, int datatype, int op,
                                void* comm, moc_stream_t stream);
int moc_train_steps_dp(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws,
                       const int64_t* labels, int slide0, int n, uint32_t use_bits,
                       float* grad_flat, int64_t grad_count, moc_allreduce_fn allreduce,
                       void* comm, int world, moc_stream_t stream);

/* e, one node: the same synchronous step with the exchange INSIDE the step kernel.  Every rank
 * writes its gradient straight into its peers' receive buffers over xGMI (IPC-mapped, fine-grained
 * memory), flags them, and sums the slices in rank order before its Adam update: two launches per
 * step instead of three plus a collective, bit-identical parameters on every rank.
 *   moc_p2p_create   allocate this rank's receive buffer for n_par = 64*D + 324 floats (current device)
 *   moc_p2p_export   write moc_p2p_handle_bytes() bytes the peers need (exchange them out of band,
 *                    e.g. torch.distributed.all_gather_object)
 *   moc_p2p_connect  `blobs` = the world's exports concatenated in rank order
 *   moc_p2p_allreduce  stand-alone sum of buf[0..n) over the ranks (self-check / small vectors)
 *   moc_p2p_error    0, or 1 + the rank that stayed silent past the time-out (5 s; MOC_P2P_TIMEOUT_MS)
 * Every rank must issue the same sequence of exchanges.  Ranks must be on one node, world <= 8. */
typedef struct moc_p2p moc_p2p_t;
int moc_p2p_handle_bytes(void);
int moc_p2p_create(int world, int rank, int64_t n_par, moc_p2p_t** out);
int moc_p2p_export(moc_p2p_t* comm, void* blob);
int moc_p2p_connect(moc_p2p_t* comm, const void* blobs);
int moc_p2p_allreduce(moc_p2p_t* comm, float* buf, int64_t n, moc_stream_t stream);
int moc_p2p_error(moc_p2p_t* comm);
int moc_p2p_destroy(moc_p2p_t* comm);
/* 1 when moc_train_steps_p2p handles these run constants (a function of them alone, so that every
 * rank takes the same decision whatever its bag sizes) */
int moc_p2p_step_supported(int C, int topk, int D, int topj);
int moc_train_steps_p2p(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws,
                        const int64_t* labels, int slide0, int n, uint32_t use_bits,
                        moc_p2p_t* comm, moc_stream_t stream);

/* f4: gated-attention MIL pooling, the aggregation of the ABMIL / CLAM baselines
 * (models/model_clam.py:41-64 Attn_Net_Gated; :178-183, :206 CLAM_SB; :291-296, :318 CLAM_MB):
 *   a = tanh(h Wa^T + ba), b = sigmoid(h Wb^T + bb)        [N, D]   (nn.Linear layout: Wa, Wb are [D, L])
 *   A_raw[k][n] = Wc[k] . (a[n] * b[n]) + bc[k]             [K, N]   -- written (the reference returns it)
 *   M[k] = sum_n softmax_n(A_raw[k])[n] * h[n]              [K, L]
 * h: device fp32 [N, L] row-major, 16-byte aligned; L % 16 == 0; D in {128, 256, 384}; K <= 64.
 * workspace: moc_gated_attention_workspace(N, L, D, K) bytes of device memory, 16-byte aligned. */
size_t moc_gated_attention_workspace(int64_t N, int L, int D, int K);
int moc_gated_attention_pool(const float* h, int64_t N, int L, const float* Wa, const float* ba,
                             const float* Wb, const float* bb, int D, const float* Wc, const float* bc,
                             int K, float* A_raw, float* M, void* workspace, size_t workspace_bytes,
                             moc_stream_t stream);

/* The backward of moc_gated_attention_pool (what autograd does behind Attn_Net_Gated.forward + softmax + mm:
 * models/model_clam.py:58-63, :178-183, :206; the CLAM trainer calls loss.backward(), utils/core_utils.py:291).
 * Given gA [K, N] (gradient at A_raw; nullable) and gM [K, L] (gradient at M; nullable; 16-byte aligned), one
 * recompute pass over the bag (the forward's main loop: no activations were kept) writes
 *   dab [N, S]        S = moc_gated_attention_dab_stride(D, K) = 2 D + 16 ceil(K / 16): columns [0, D) and [D, 2 D)
 *                     the gradients at the two pre-activations h Wa^T + ba | h Wb^T + bb; column 2 D + k the
 *                     softmax p[k][n] = softmax_n(A_raw[k]); the rest zero
 *   ds  [K, N]        total gradient at A_raw: p (dp - sum_n p dp) + gA with dp[k][n] = h[n] . gM[k]
 *   dcol[(2 + K) D]   d_ba [D] | d_bb [D] | d_Wc [K, D]
 *   dbc [K]           d_bc
 * (every sum in a fixed order: deterministic).  What remains are two plain GEMMs the caller hands to a library
 * (rocBLAS / hipBLASLt: torch.mm) over X = [Wa; Wb; gM; 0] ([S, L]; zeros for gM when it is null):
 *   dh = dab X      and      dab^T h = [dWa; dWb; (M, unused)].
 * Same shape limits as the forward; workspace: moc_gated_attention_backward_workspace bytes, 16-byte aligned. */
size_t moc_gated_attention_backward_workspace(int64_t N, int L, int D, int K);
int moc_gated_attention_dab_stride(int D, int K);
int moc_gated_attention_backward(const float* h, int64_t N, int L, const float* Wa, const float* ba,
                                 const float* Wb, const float* bb, int D, const float* Wc, int K,
                                 const float* A_raw, const float* gA /*nullable*/, const float* gM /*nullable*/,
                                 float* dab, float* ds, float* dcol, float* dbc, void* workspace,
                                 size_t workspace_bytes, moc_stream_t stream);

/* a10-a15 fused: `n` consecutive meta-steps (one slide each, slides slide0..slide0+n-1 in
 * order, one Adam step per slide: main_moc.py:380-410), parameters and Adam moments
 * updated in place.  Per-slide loss/pooled land in ws->loss / ws->pooled. */
int moc_train_steps(const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws,
                    const int64_t* labels, int slide0, int n, uint32_t use_bits,
                    moc_stream_t stream);

/* The same pass as ONE graph launch (main_moc.py:380-410: the whole `for data in train_loader` body, n slides).
 * moc_train_steps issues 2 n + 1 kernel launches; at the start of a run -- or of a short timed region -- the stream is
 * empty and the chain of 20-us meta-steps cannot run ahead of the host.  A `moc_step_graph_t` keeps each distinct pass
 * (work arrays, meta-learner tensors, slide range) as an instantiated hipGraph and replays it with one call.  It is the
 * second explicit handle of this ABI (after moc_p2p_t): it owns host-side state only -- the graphs, one pinned staging
 * buffer, an event; the device memory it uses is the caller's `device_ws` (moc_step_graph_workspace_bytes(max_steps):
 * a step counter + the Adam coefficients of the next max_steps steps, which the captured kernels read by position
 * because kernel arguments are frozen at capture while the bias corrections change with every step).
 *   - results are bit-identical to moc_train_steps (same kernels, same coefficient floats);
 *   - all calls on one handle must be ordered on one stream (the counter lives in stream order);
 *   - M->step is read on every call: if it is not where the handle left it (another optimizer stepped, a checkpoint
 *     was loaded), the device counter is put right first; new hyper-parameters or an exhausted table rebuild the table;
 *   - shapes outside the one-launch steps, n > max_steps, or a runtime that refuses capture / instantiation fall back
 *     to moc_train_steps (moc_step_graph_stats tells which path ran).
 * The caller advances M->step by n afterwards, as with moc_train_steps. */
typedef struct moc_step_graph moc_step_graph_t;
size_t moc_step_graph_workspace_bytes(int max_steps);
int moc_step_graph_create(void* device_ws, size_t ws_bytes, moc_step_graph_t** out);
int moc_step_graph_destroy(moc_step_graph_t* G);
int moc_step_graph_stats(const moc_step_graph_t* G, int* captures, int* replays, int* eager_calls);
int moc_train_steps_graph(moc_step_graph_t* G, const moc_batch_t* B, const moc_meta_t* M, const moc_meta_ws_t* ws,
                          const int64_t* labels, int slide0, int n, uint32_t use_bits, moc_stream_t stream);

/* ---- compute units kept free of the score pass (round 3) -----------------------------------------------------------
 * Phase A of the NEXT pass (main_moc.py:322-375 for every slide: no trainable parameter) is issued a pass ahead on a
 * side stream while the sequential meta-steps of this pass (main_moc.py:380-410) run.  The score pass fills every CU
 * with persistent workgroups, and a 16-wave meta-step workgroup fits on no CU that holds one of them: the meta-steps
 * stalled for the length of the score pass.  (Confining the side stream's queue with a CU mask,
 * hipExtStreamCreateWithCUMask, was measured: the masked queue is not scheduled beside the unmasked one at all -- the
 * rate halves; profiles/NOTES.md.)  Instead the score pass itself stays off a set of compute units:
 *   moc_batch_t.tile_ticket  (device int32[MOC_TICKET_WORDS], nullable): the score pass hands out its 16-row tiles
 *       through MOC_TICKET_QUEUES counters (eight per XCD, 256 bytes apart, two tiles per ticket; a wave whose own has
 *       run out takes from any other; eight more words rank the workgroups of each XCD) instead of a static stride per
 *       wave, so any number of its workgroups can do all the work (moc_scores zeroes them in stream order before each
 *       launch);
 *   moc_batch_t.cu_reserved  (device uint32[128], nullable; needs tile_ticket): bit (xcc * 256 + HW_ID[15:8]) set = a
 *       workgroup of the score pass that finds itself on that compute unit ends at once (the first eight workgroups of
 *       a launch never do: progress does not depend on the table being right).
 * moc_cu_census finds out which (xcc, HW_ID[15:8]) slots the device has: it launches `n_wg` one-wave workgroups that
 * each hold their CU for ~`hold_us` microseconds and counts them in hist[xcc * 256 + HW_ID[15:8]] (device int32
 * [16 * 256], zeroed by the caller).  The host picks the reserved slots from it (moc_amd/engine.py: reserved_cus). */
int moc_cu_census(int32_t* hist, int n_wg, int hold_us, moc_stream_t stream);

/* ---- generic pooling / ranking (a12, a17 and the index_* helpers) ----------
 * For each segment s (rows seg_off[s] .. seg_off[s+1]) and class c: rank rows by
 * keys[c*key_stride + row] (largest first, or smallest first when `smallest`),
 * take k = min(K, len) and write mean(vals[c*val_stride + row]) over them to
 * pooled[s*C + c] (utils/patch_selection_classifier.py:18-32, :35-80, :127-171).
 * idx_out (nullable, int32 [n_seg, C, K]): chosen rows relative to the segment,
 * ordered by key (ties: lower row first); cnt_out (nullable) [n_seg, C].
 * K <= 4096. */
int moc_topk_mean(const float* keys, int64_t key_stride, const float* vals, int64_t val_stride,
                  const int64_t* seg_off, const int32_t* seg_len /*nullable*/, int n_seg, int C, int K,
                  int smallest, float* pooled, int32_t* idx_out, int32_t* cnt_out,
                  moc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MOC_HIP_H */
