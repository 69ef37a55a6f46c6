/*
 * emx.h -- C ABI of libemx.so: the hand-written HIP (gfx950 / MI355X) kernels behind
 * encodermap_amd's training hot path.
 *
 * The reference (AG-Peter/encodermap, pure Python on TensorFlow) has no FFI boundary of its own: every
 * function below replaces a group of TensorFlow op call sites, cited as file:line relative to the upstream
 * repository @ 2025-05-23.  INTEGRATION.md shows the ctypes stub that binds each symbol.
 *
 * Contract (all functions):
 *   - every pointer is a DEVICE pointer owned by the caller (row-major float32 unless stated), except
 *     `const float sig[6]` / scalars, which are host values copied at launch;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it, nothing
 *     synchronises, allocates or frees, so every entry point is legal inside a hipGraph capture (one exception: the FIRST call
 *     that hands the library a set of counter words must run outside a capture, see "counter words");
 *   - return value 0 = launched, negative = rejected before any launch (bad argument, unsupported shape);
 *     emx_last_error() returns the thread-local message of the last rejection;
 *   - re-entrant; the only global mutable state are the launch count and the note of which counter words were zeroed (mutex); one host thread per GPU/rank is the intended use;
 *   - results are deterministic run to run: no float atomics, fixed-order two-pass reductions.
 *
 * Row-block arguments (`row_begin`, `row_end`) implement the data-parallel partition of SURVEY.md section
 * 8e: a rank evaluates only the terms owned by rows [row_begin,row_end) of the GLOBAL batch of `B` rows,
 * with the global 1/B (or 1/B^2) normalisation, so that a SUM all-reduce over ranks yields the
 * single-process value.
 */
#ifndef EMX_H
#define EMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMX_VERSION 1
/* scratch floats every two-pass scalar reduction needs (per call, may be shared by calls ordered on one stream) */
#define EMX_REDUCE_WS 1024

typedef void* emx_stream_t; /* hipStream_t */

/* cost variants of loss_functions.py:598-613 */
enum { EMX_MEAN_ABS = 0, EMX_MEAN_SQUARE = 1, EMX_MEAN_NORM = 2 };

/* error codes */
enum { EMX_OK = 0, EMX_EINVAL = -1, EMX_EUNSUPPORTED = -2, EMX_ELAUNCH = -3, EMX_EWORKSPACE = -4 };

int emx_version(void);
const char* emx_last_error(void);
/* name of the device the library was built for ("gfx950"); used by the host side to fail loudly */
const char* emx_target_arch(void);
/* number of kernel launches this library has issued in the calling process so far (every entry point counts the launches it
 * makes, captured into a hipGraph or not): the difference across one training step is the step's launch count -- what a
 * launch-bound configuration is priced by (bench.py: launches_per_step; the reference's train_step, models.py:3367-3401, is
 * ~10^2 TensorFlow op launches) */
int64_t emx_launch_count(void);
/* Counter words.  Some launches hand work from workgroup to workgroup INSIDE the launch (arrival counts, a sticky give-up flag):
 * the `sync` arguments below.  They must be zero before
 * the first such launch and every launch leaves them at zero (the give-up flag excepted).  The library does not rely on the caller
 * for that: the first time an entry point is handed such a pointer -- per process -- it zeroes the words itself on the caller's
 * stream, in front of its launch.  That first sight must not happen inside a stream capture (EMX_EINVAL: the memset would become a
 * graph node and clear the give-up flag on every replay) -- run the call once eagerly first, as a warm-up step does.  The note is
 * by ADDRESS: a caller that frees such memory and gets the same address back from its allocator (or recycles it for something
 * else) must hand it over zeroed -- allocate these few words zeroed, as you would a lock -- or call emx_forget_counters(ptr)
 * (NULL: forget all), after which the next call zeroes it again.
 * The same call releases what the library keeps PER SCRATCH ADDRESS: the ticket words of the one-launch split reduction
 * (emx_sketchmap_tiles) live in library-owned device memory, one small block per workspace address, allocated with hipMalloc at the
 * address's first sight -- hence not inside a capture either, and not while another thread captures in global mode.
 * emx_forget_counters(workspace) hands that block back to the library's pool; the next call with the address gets a block zeroed
 * anew.  Call it when a workspace is freed, and after a launch was ABORTED (device reset, a give-up) with tickets taken: that is
 * how a caller re-zeroes them.  It is the caller's statement that NOTHING addresses the block any more: drop every hipGraph that was
 * captured with a launch on that workspace first, and let launches in flight on it finish -- the block may be handed to another
 * workspace address afterwards.  Blocks the library takes away on its own (a later call on the same workspace that needs more
 * ticket words; emx_forget_counters(NULL)) are never handed out again: graphs captured before stay valid. */
void emx_forget_counters(void* words);

/* ------------------------------------------------------------------------------------------------ *
 * K1/K2  sketch-map sigmoid pairwise-distance cost, forward + gradient wrt the latent
 *   replaces: pairwise_dist            encodermap/misc/distances.py:179-255
 *             pairwise_dist_periodic   encodermap/misc/distances.py:144-176
 *             sigmoid                  encodermap/misc/distances.py:66-88
 *             sigmoid_loss_func        encodermap/loss_functions/loss_functions.py:351-367
 *             distance_loss_func       encodermap/loss_functions/loss_functions.py:266-296 (the x scale)
 *             center_loss_func         encodermap/loss_functions/loss_functions.py:422-449 (optional fused term)
 *   X [B, D] (row stride ldx floats) high-d rows (data, no gradient); Z [B, L] latent, any L >= 1 (widths beyond 8 are walked eight columns at a time).
 *   periodicity: INFINITY -> Euclidean, else the period of X.
 *   data_span: for periodic X, max - min over the data the rows are drawn from if the caller knows it, else 0.  Below 1.5
 *   periods every |x_i - x_j| < 1.5 p, where the reference's min(|d|, p - |d|)^2 is the squared wrapped difference, and the
 *   tile pass runs on wrapping 32-bit phases; 0 (unknown), NaN or a wider span keep the float form, which follows the
 *   reference's formula for any |d| (its values go negative beyond one period and are squared).  Ignored for Euclidean X.
 *   sig = (sig_h, a_h, b_h, sig_l, a_l, b_l).
 *   *dist_loss   = scale / B^2 * sum_{i in rows} sum_{j<B} (S_h(dh_ij) - S_l(dl_ij))^2
 *   *center_loss = center_scale / (B*L) * sum_{i in rows} sum_l z_il^2      (skipped if center_loss == NULL)
 *   dZ[i,:] (rows in the block only) = d(dist + center)/dz_i of the FULL-batch loss; pairs with dl == 0
 *   contribute zero gradient (the reference's mask, distances.py:247-253).
 *   ws: scratch of at least emx_sketchmap_ws_bytes(B, L) bytes; its contents need no initialisation and mean nothing between calls
 *   of the tiles / finish pair.  (Small batches of long rows split a tile's reduction over workgroups of ONE launch: the last of a
 *   tile's workgroups adds their shares and runs the epilogue.  Their ticket words live in memory of the LIBRARY's, one zeroed
 *   block per workspace address it has seen -- so the first such call for a workspace must run outside a stream capture.)
 * ------------------------------------------------------------------------------------------------ */
size_t emx_sketchmap_ws_bytes(int B, int L);
int emx_sketchmap_fwd_bwd(const float* X, int B, int D, int ldx, const float* Z, int L,
                          float periodicity, float data_span, const float sig[6], float scale,
                          float center_scale, int row_begin, int row_end,
                          float* dist_loss, float* center_loss, float* dZ,
                          void* ws, size_t ws_bytes, emx_stream_t stream);

/* The two launches of emx_sketchmap_fwd_bwd as separate entry points (per-kernel timing in bench.py, and for
 * callers that want other work between the O(B^2 D) tile pass and the O(B) reduction).  Same arguments; the
 * tile pass leaves per-tile partial sums in `ws`, the finish pass reduces them (scale, center cost, dZ). */
int emx_sketchmap_tiles(const float* X, int B, int D, int ldx, const float* Z, int L, float periodicity, float data_span,
                        const float sig[6], int row_begin, int row_end, void* ws, size_t ws_bytes,
                        void* split_ws /* may be NULL; Euclidean distances only */, emx_stream_t stream);
/* Split rows for emx_sketchmap_tiles: x - x_0 (row 0 of the batch) as two f16 planes with a power-of-two scale per row
 * (h = f16(s x), l = f16(s x - h): 22 bits), plus the squared row norms and 1/s.  With them the tile pass forms its Gram
 * products on the f16 matrix cores (three v_mfma_f32_32x32x16_f16 per 16 features -- h.l, l.h, h.h, every partial product
 * exact in the f32 accumulator: the accuracy class of the f32 matrix instruction at a fifth of its cycles, and -- unlike
 * that instruction -- without holding the VALU that the sigmoid epilogues of the other waves need).  emx_gather_rows and emx_next_batch produce them beside the gather;
 * emx_sketchmap_split makes them from a dense [B, D] block.  split_ws: emx_sketchmap_split_bytes(B, D) bytes, 16-byte aligned. */
size_t emx_sketchmap_split_bytes(int B, int D);
int emx_sketchmap_split(const float* X, int B, int D, int ldx, void* split_ws, emx_stream_t stream);
int emx_sketchmap_finish(int B, int D, const float* Z, int L, float periodicity /* as given to emx_sketchmap_tiles: it selects the
                         tile size, hence the layout of ws */, const float sig[6], float scale, float center_scale,
                         int row_begin, int row_end, float* dist_loss, float* center_loss, float* dZ,
                         void* ws, size_t ws_bytes, emx_stream_t stream);

/* Where emx_sketchmap_tiles leaves its results, for callers that fold the O(B) reduction into kernels they launch anyway
 * (emx_mlp_backward adds the latent gradient while it forms delta; emx_finish_step sums the cost):
 *   d cost / d z[row, c] = gscale * sum_{s < n_slots} partials[s * slot_stride + (row - row_base) * L + c] + center_grad * z[row, c]
 *   distance cost = loss_scale * sum(tile_loss[0 .. n_tiles));  center cost = center_loss_scale * sum_{own rows} z^2 */
typedef struct {
  const float* partials;
  const float* Z;
  int64_t slot_stride;
  int n_slots;
  int row_base;
  float gscale;
  float center_grad;
} emx_latent_grad;
typedef struct {
  emx_latent_grad grad;
  const float* tile_loss;
  int n_tiles;
  float loss_scale;
  float center_loss_scale;
} emx_sketchmap_partials;
int emx_sketchmap_describe(int B, int D, int ldx, const float* X, const float* Z, int L, float periodicity, const float sig[6],
                           float scale, float center_scale, int row_begin, int row_end, void* ws, size_t ws_bytes,
                           emx_sketchmap_partials* out);

/* Data parallel with the objective of the single process (loss_functions.py:351-367 needs ALL B^2 pairs of
 * the global batch): the symmetric tile pass over the whole batch with its upper-triangle tiles DEALT to the ranks -- tile t
 * (row-major triangle index) belongs to rank t % world, so every rank does 1 / world of the pair work and keeps row and column
 * partials of its tiles; cost slots of other ranks' tiles are zeroed (sum(tile_loss) is the rank's share of the cost).
 * emx_sketchmap_reduce_dealt then forms, for ALL rows, d cost / d z from this rank's tiles (+ the center cost's gradient on
 * rows [own_begin, own_end)): the SUM over ranks -- one all-reduce of B x L floats -- is the gradient emx_sketchmap_finish
 * would give.  Same workspace as emx_sketchmap_tiles(row_begin = 0, row_end = B). */
int emx_sketchmap_tiles_dealt(const float* X, int B, int D, int ldx, const float* Z, int L, float periodicity, float data_span,
                              const float sig[6], void* ws, size_t ws_bytes, void* split_ws /* may be NULL */, int rank, int world,
                              emx_stream_t stream);
int emx_sketchmap_reduce_dealt(int B, int D, const float* Z, int L, float periodicity, const float sig[6], float scale, float center_scale,
                               int own_begin, int own_end, int rank, int world, float* dZ, void* ws, size_t ws_bytes, emx_stream_t stream);

/* materialising variants, for parity tests of a2/a3 only (out is [B,B]) */
int emx_pairwise_dist(const float* X, int B, int D, int ldx, float periodicity, float* out, emx_stream_t stream);

/* ------------------------------------------------------------------------------------------------ *
 * K13 + K4  data feed: gather a mini-batch by row index and (optionally) build the sin/cos input
 *   replaces: tf.data from_tensor_slices/shuffle/repeat/batch   encodermap/autoencoder/autoencoder.py:845-856
 *             SequentialModel.encoder sin/cos concat             encodermap/models/models.py:3347-3351
 *             PeriodicInput.call                                 encodermap/models/layers.py:204-215
 *   data [N, D]; idx int64 row numbers, batch number c occupies idx[c*B .. c*B+B); `cursor` (device int64,
 *   may be NULL = 0) selects the batch so that a captured graph walks a pre-drawn index block without host
 *   work (emx_step_tick advances it); xb [B, D] gathered rows (may be NULL);
 *   xin [B, 2D] = [sin(x*2pi/p) | cos(x*2pi/p)] (may be NULL; periodicity ignored then).
 * ------------------------------------------------------------------------------------------------ */
/* The shuffle of the data feed: out[i] = pi(i), i < n, a pseudo-random permutation of 0 .. n-1 chosen by (seed, epoch) --
 * one pass of Dataset.shuffle(N, reshuffle_each_iteration=True) (autoencoder.py:845-856; the reference's own order comes
 * from TensorFlow's generator and is not reproducible elsewhere).  pi is a keyed bijection, evaluated independently per
 * element: a 6-round Feistel network on the ceil(log2 n) bits of the index (round function: the 32-bit murmur3 finaliser
 * of the half-word xor a round key), cycle-walked until the value falls below n -- no sort, no atomics, no global
 * synchronisation, and bit-identical to its host statement encodermap_amd.layout.feistel_permutation. */
int emx_permute(int64_t* out, int64_t n, uint64_t seed, uint64_t epoch, emx_stream_t stream);

int emx_gather_rows(const float* data, int64_t N, int D, const int64_t* idx, const int64_t* cursor, int B,
                    float periodicity, float* xb, float* xin, void* split_ws /* may be NULL: emx_sketchmap_split of the gathered rows */,
                    emx_stream_t stream);
/* PeriodicInput on an already dense [B, D] block (encode()) */
int emx_periodic_in(const float* x, int B, int D, int ldx, float periodicity, float* out, int ldo, emx_stream_t stream);
/* PeriodicOutput: out[b, k] = atan2(y[b, off_sin+k], y[b, off_cos+k]) * p/2pi  (models.py:3353-3359, layers.py:1118-1125) */
int emx_periodic_out(const float* y, int B, int n, int ldy, int off_sin, int off_cos, float periodicity,
                     float* out, int ldo, emx_stream_t stream);

/* ------------------------------------------------------------------------------------------------ *
 * K5  row-mean cost on (periodic) differences, forward + gradient wrt the network output
 *   replaces: periodic_distance + mean_abs/mean_square/mean_norm  encodermap/misc/distances.py:113-141,
 *             auto_loss_func / dihedral_loss_func / angle_loss_func / side_dihedral_loss_func
 *             encodermap/loss_functions/loss_functions.py:586-626, 671-706, 832-868, 751-788
 *             and, fused, PeriodicOutput's atan2 with its backward (models.py:3353-3359)
 *   target [B, n] (stride ldt).  y (stride ldy): if periodic_output != 0 the prediction is
 *   atan2(y[:,off_sin+k], y[:,off_cos+k]) * p/2pi and dY receives the gradient for both halves; otherwise
 *   the prediction is y[:, off_sin+k] itself.  periodicity = INFINITY -> plain |pred - target|.
 *   *loss = coef * variant(delta) over rows [row_begin,row_end) normalised by the GLOBAL B;
 *   dY (same layout/stride as y) is WRITTEN for the addressed columns of rows in the block (NULL: forward only).
 *   loss == NULL: no final sum is launched; ws[0 .. EMX_REDUCE_WS) then holds per-workgroup partials that already carry
 *   coef (and *coef_scale) and add up to the cost (emx_finish_step sums them).
 *   `coef_scale` (device float, e.g. the soft-start scale of callbacks.py:584-606) multiplies coef on the device so
 *   that a captured graph follows the step counter; `dpred_add` lets the back-mapping path (cartesian cost) join
 *   the gradient before the atan2 backward.  The same entry point serves cartesian_loss
 *   (loss_functions.py:1020-1065): periodic_output = 0, periodicity = INFINITY, y = out_pair, target = inp_pair.
 * ------------------------------------------------------------------------------------------------ */
int emx_periodic_cost_fwd_bwd(const float* y, int ldy, int off_sin, int off_cos, int periodic_output,
                              const float* target, int ldt, int B, int n, float periodicity, int variant,
                              float coef, const float* coef_scale /* device scalar multiplied into coef, may be NULL */,
                              const float* dpred_add /* [B, n] gradient wrt the prediction from other paths, may be NULL */,
                              int row_begin, int row_end, float* loss, float* dY,
                              float* ws /* >= EMX_REDUCE_WS floats */, emx_stream_t stream);

/* Several row-mean costs of one step in ONE launch (the ADC model evaluates a dihedral, an angle and a side-dihedral cost on
 * column windows of the same output rows, models.py:2260-2521): each job is the argument list of
 * emx_periodic_cost_fwd_bwd with loss == NULL -- its ws[0 .. EMX_REDUCE_WS) receives the per-workgroup partials -- over the
 * common rows [row_begin, row_end) of a batch of B.  The same numbers as n_jobs separate calls.  Jobs must not share ws. */
typedef struct {
  const float* y;
  int ldy, off_sin, off_cos, periodic_output;
  const float* target;
  int ldt, n;
  float periodicity;
  int variant;
  float coef;
  const float* coef_scale; /* may be NULL */
  const float* dpred_add;  /* may be NULL */
  float* dY;               /* may be NULL */
  float* ws;
} emx_cost_job;
#define EMX_MAX_COST_JOBS 4
int emx_periodic_cost_batch(const emx_cost_job* jobs, int n_jobs, int B, int row_begin, int row_end, emx_stream_t stream);

/* center cost alone: mean(z^2)*scale over the global batch (loss_functions.py:422-449); dZ written or accumulated */
int emx_center_cost_fwd_bwd(const float* Z, int B, int L, float scale, int row_begin, int row_end,
                            float* loss, float* dZ, int accumulate, emx_stream_t stream);

/* description of the dense network (see K12 below); declared here because the optimizer keeps the operand store of the
 * dense-layer kernels current */
#define EMX_MAX_LAYERS 12   /* Dense layers of ONE description: n_neurons of up to six entries (encoder + mirrored decoder).  A deeper
                             * network is run as WINDOWS of consecutive layers: one description per window over the same parameter
                             * buffer (w_off / b_off are offsets into it), each with its own operand store and fragment store;
                             * emx_mlp_forward chains them through h, emx_mlp_backward through dbelow -> dtop, emx_mlp_wgrad writes a
                             * window's part of the one gradient / partials buffer, the optimizer runs without a store and
                             * emx_mlp_mirror rebuilds each window's afterwards (host side: encodermap_amd/_lib.py, MlpPlan.windows) */
#ifndef EMX_WGRAD_SLICE_ROWS
#define EMX_WGRAD_SLICE_ROWS 128   /* rows per weight-gradient slice; doubled until ... */
#endif
#ifndef EMX_WGRAD_MAX_SLICES
#define EMX_WGRAD_MAX_SLICES 32    /* ... at most this many slices remain (emx_mlp_wgrad_slices) */
#endif
typedef struct {
  int n_layers;
  int width[EMX_MAX_LAYERS + 1];
  int act[EMX_MAX_LAYERS];
  int64_t w_off[EMX_MAX_LAYERS];
  int64_t b_off[EMX_MAX_LAYERS];
} emx_mlp_desc;

/* The rows of the NEXT batch, gathered beside the optimizer (the same rows emx_gather_rows would produce for batch
 * *cursor + 1, skipped when that is >= n_batches): the optimizer is a short latency-bound stream over ~1e5 parameters, the
 * gather runs on other CUs meanwhile, and the next step starts with its rows already in xb / xin. */
typedef struct {
  const float* data;      /* [n_rows, D] resident data set */
  int64_t n_rows;
  int D;
  const int64_t* idx;     /* gather index block, n_batches x B */
  int n_batches;
  const int64_t* cursor;  /* device scalar: the CURRENT batch */
  int B;
  float periodicity;      /* used when xin != NULL */
  float* xb;              /* may be NULL: [B, D] raw rows */
  float* xin;             /* may be NULL: [B, 2 D] = [sin | cos] */
  void* split_ws;         /* may be NULL: emx_sketchmap_split_bytes(B, D) bytes, the split rows of emx_sketchmap_split */
  const void* adc_head;   /* may be NULL.  Else: the head of the next ADC step instead of the row gather -- the device record written
                             by emx_adc_head_describe (what emx_adc_batch_head would do for batch *cursor + 1); of the fields above
                             idx, n_batches, cursor, B are used, D = the feature columns whose batch means are wanted (0: none) */
} emx_next_batch;

/* ------------------------------------------------------------------------------------------------ *
 * K11 + K10  fused clip -> Adam on the flat parameter buffer, L2 regulariser folded in
 *   replaces: tf.keras.optimizers.Adam(lr, clipvalue=1.0).apply_gradients  encodermap/autoencoder/autoencoder.py:741-743,
 *             encodermap/models/models.py:3395 ; regularization_loss_func   loss_functions.py:488-506
 *   g_total = g + l2coef[i]*w[i]  (l2coef = 2*lambda on kernel elements, 0 on biases; may be NULL)
 *   g_total = clamp(g_total, -clip, clip)  (clip <= 0 disables)
 *   m += (g-m)(1-b1); v += (g^2-v)(1-b2); w -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps),  t = *step + 1
 *   *reg_loss = sum_i 0.5*l2coef[i]*w[i]^2 evaluated on the weights BEFORE the update (may be NULL: then ws[0 ..
 *   EMX_REDUCE_WS) holds partials that add up to it).
 *   `step` is a device int64 holding the number of completed updates; it is NOT modified here (emx_step_tick does).
 * ------------------------------------------------------------------------------------------------ */
int emx_adam_step(float* w, const float* g, float* m, float* v, const float* l2coef, int64_t n,
                  float lr, float beta1, float beta2, float eps, float clip,
                  const int64_t* step, float* reg_loss, float* ws /* >= EMX_REDUCE_WS floats */,
                  const emx_mlp_desc* mirror_desc /* may be NULL */, float* mirror /* may be NULL: operand store (emx_mlp_mirror) rewritten
                  with the new kernels */, const emx_next_batch* next /* may be NULL */,
                  int bookkeeping_done /* see below */, emx_stream_t stream);

/* emx_adam_step on the weight-gradient slice partials left by emx_mlp_wgrad (gradient = fixed-order sum over n_slices slices of
 * g_partials[s * partial_stride + i]); no separate reduction pass.  ws[0 .. EMX_REDUCE_WS) receives partials of the regulariser
 * cost (summed by emx_finish_step). */
int emx_adam_step_slices(float* w, const float* g_partials, int n_slices, int64_t partial_stride, float* m, float* v, const float* l2coef,
                         int64_t n, float lr, float beta1, float beta2, float eps, float clip, const int64_t* step, float* ws,
                         const emx_mlp_desc* mirror_desc, float* mirror, const emx_next_batch* next, int bookkeeping_done,
                         emx_stream_t stream);
/* bookkeeping_done != 0: the step's emx_finish_step has ALREADY run when the optimizer launches (it rode along with
 * emx_mlp_wgrad, emx_finish_desc), so *step already counts this update (t = *step), the batch under *cursor already is the
 * next one, and ws receives the regulariser shares of the NEW weights -- the regulariser term of the NEXT step's cost
 * (0.5 sum l2coef w^2 is a function of the weights the next forward pass uses).  emx_l2_partials provides those shares
 * for the weights a run starts from. */
int emx_l2_partials(const float* w, const float* l2coef, int64_t n, float* ws /* EMX_REDUCE_WS floats */, emx_stream_t stream);

/* The optimizer launch of a DATA-PARALLEL step: emx_adam_step with bookkeeping_done = 1 on the
 * all-reduced gradient g, plus one extra workgroup that writes the step's history row from the all-reduced cost terms:
 *   history[*hstep - 1] = sum(terms[0 .. n_terms)),  term_history[(*hstep - 1) * n_terms + i] = terms[i]
 * (*hstep was advanced by the emx_finish_desc that rode along with emx_mlp_wgrad, in front of the all-reduce, together with
 * *step and *cursor; the row needs the REDUCED terms, so it is written here, behind it -- the Keras History entry of
 * models.py:3399-3401).  No launch of its own for the scalar epilogue: a rank's step is five launches + one slice reduction
 * + the collective.  history / term_history may be NULL. */
int emx_adam_step_history(float* w, const float* g, float* m, float* v, const float* l2coef, int64_t n, float lr, float beta1, float beta2,
                          float eps, float clip, const int64_t* step, float* ws, const emx_mlp_desc* mirror_desc, float* mirror,
                          const emx_next_batch* next, const float* terms, int n_terms, float* history, float* term_history,
                          int64_t capacity, const int64_t* hstep, emx_stream_t stream);

/* emx_adam_step + emx_step_tick behind one entry point, optionally consuming the weight-gradient slice partials of emx_mlp_wgrad_partials directly
 * (gradient = fixed-order sum over n_slices slices of g_partials[s * partial_stride + i]) so that no separate
 * reduction pass is needed on a single GPU.  terms[reg_slot] receives the regulariser cost; then
 * history[*hstep] = sum(terms), term_history row, ++*hstep, ++*cursor (may be NULL), ++*opt_step.
 * ws: EMX_REDUCE_WS floats. */
int emx_adam_tick(float* w, const float* g, const float* g_partials, int n_slices, int64_t partial_stride, float* m, float* v,
                  const float* l2coef, int64_t n, float lr, float beta1, float beta2, float eps, float clip, int64_t* opt_step,
                  float* terms, int n_terms, int reg_slot, float* history, float* term_history, int64_t capacity, int64_t* hstep,
                  int64_t* cursor, float* ws, const emx_mlp_desc* mirror_desc, float* mirror, emx_stream_t stream);

/* Scalar epilogue of a step in ONE single-workgroup launch: every job adds `n` floats (or their squares) in a fixed order
 * and stores scale * (*dev_scale) * sum into terms[slot]; with hstep != NULL the bookkeeping of emx_step_tick follows
 * (history[*hstep] = sum(terms), ++counters).  Replaces the per-cost final-sum launches. */
typedef struct {
  const float* src;
  int64_t n;
  float scale;
  const float* dev_scale; /* may be NULL */
  int slot;
  int square;
} emx_sum_job;
#define EMX_MAX_SUM_JOBS 8
typedef struct {   /* the arguments of emx_finish_step, for launches that carry it along (emx_mlp_wgrad) */
  const emx_sum_job* jobs;
  int n_jobs;
  float* terms;
  int n_terms;
  float* history;
  float* term_history;
  int64_t capacity;
  int64_t* hstep;
  int64_t* cursor;
  int64_t* counter2;
} emx_finish_desc;
int emx_finish_step(const emx_sum_job* jobs, int n_jobs, float* terms, int n_terms, float* history, float* term_history,
                    int64_t capacity, int64_t* hstep, int64_t* cursor, int64_t* counter2, emx_stream_t stream);

/* end-of-step bookkeeping on the device (so a captured graph can be replayed without host round trips):
 * history[*step] = sum(terms[0..n_terms)), term_history[*step, :] = terms (if not NULL), then ++*step
 * (and ++*cursor, the batch cursor of emx_gather_rows).
 * replaces the Keras History / _my_train_counter bookkeeping, encodermap/models/models.py:3399-3401 */
int emx_step_tick(const float* terms, int n_terms, float* history, float* term_history, int64_t capacity,
                  int64_t* step, int64_t* cursor /* may be NULL; ++*cursor */,
                  int64_t* counter2 /* may be NULL; ++*counter2: the optimizer's update counter */, emx_stream_t stream);

/* ------------------------------------------------------------------------------------------------ *
 * K12  dense encoder/decoder, fused over layers, on the 16-bit matrix cores with float32-accurate operands
 *   replaces: tf.keras.layers.Dense forward (y = act(x W + b), W [in,out])  encodermap/models/models.py:3179-3220
 *             and its autodiff (tape.gradient)                              encodermap/models/models.py:3392-3393
 *   The network is described once by an emx_mlp_desc (host struct): width[0] is the input width, width[l+1]
 *   the outputs of layer l, act[l] the layer's activation -- 0 linear, 1 tanh (the fast path), 2 relu, 3 sigmoid, 4 elu, 5 softplus,
 *   6 selu, 7 softsign, 8 exponential: the Keras activations (models.py:3179-3220 hands any Keras name to Dense) whose derivative is
 *   a function of the layer's OUTPUT, which is what the backward kernels keep --, w_off/b_off the float offsets of the [in,out] kernel
 *   and the bias inside the flat parameter buffer `params` (encodermap_amd/layout.py).
 *   h[l] / delta[l] (host arrays of device pointers, [B, width[l+1]] contiguous; an entry may be NULL: not kept) receive
 *   the activations and d loss / d pre-activation of layer l as float32.  Layer ranges let a data-parallel rank run the
 *   encoder on all rows and the decoder on its own row block.
 *   Arithmetic: every float32 operand is carried as a power-of-two-scaled pair of f16 values (22 significant bits;
 *   csrc/wop.h) and a product runs as three 16-bit matrix instructions with an f32 accumulator -- the accuracy class
 *   of an f32 fused-multiply-add chain at a fraction of the f32 matrix instruction's cycles.  A row's results do not
 *   depend on the row range or batch it is computed in.
 * ------------------------------------------------------------------------------------------------ */

/* Weight operand store ("mirror"): a second buffer holding, per layer, the kernel as block-scaled f16 pairs in the order
 * the row-block kernels stream it -- for the forward pass (B[k = in][col = out]) and for the backward-data pass
 * (B[k = out][col = in]) -- as 1-KB fragments of v_mfma_f32_16x16x32_f16 plus one scale per 32 x 32 block (layout:
 * csrc/wop.h).  emx_mlp_mirror_floats floats; emx_mlp_mirror builds it from `params`, emx_adam_step* keep it current when
 * they are handed the descriptor. */
size_t emx_mlp_mirror_floats(const emx_mlp_desc* desc);
int emx_mlp_mirror(const emx_mlp_desc* desc, const float* params, float* mirror, emx_stream_t stream);

/* Weight-gradient operand store ("aop"): what the forward / backward launches leave for emx_mlp_wgrad -- the network
 * input, every h_l and every delta_l as block-scaled f16 fragments of v_mfma_f32_32x32x16_f16 with the reduction index
 * on the batch rows (layout: csrc/aop.h).  emx_mlp_aop_bytes(desc, n_rows) bytes for batches of up to n_rows rows; the
 * same `aop_rows` = n_rows must be passed to every launch that uses the store.  aop == NULL: inference, nothing is
 * kept.  With a store the row range must start at a multiple of 16. */
size_t emx_mlp_aop_bytes(const emx_mlp_desc* desc, int n_rows);

/* layers [layer_begin, layer_end) on rows [row_begin,row_end); `in` [*, width[layer_begin]] (row stride ld_in).
 * h[l] (and delta[l] of the backward launches below) are tables of row pointers, one per layer of the description: a NULL entry is a
 * tensor the caller does not keep -- its float32 rows are not written (the launch still forms them on chip and leaves their
 * fragments in `aop`).  Entries a launch READS must be there: emx_mlp_backward needs h[l] of every layer with an activation in its
 * range, the inference result is h[layer_end - 1].
 * Layer widths.  The row-block kernels keep a layer's source rows on chip (LDS), which bounds a layer they run in ONE launch to
 * ~2400 source columns.  The reference has no such bound (models/models.py:3133-3160: Dense layers of any width): a range beyond it
 * runs, behind these same two entry points, as SEGMENTS -- the longest runs of layers one launch takes, and single layers whose
 * source is streamed (csrc/wide.h) -- with the same arithmetic and the same fragments left in `aop`.  Between two segments the
 * float32 rows travel through HBM: h[l] of the layers at the cuts (forward), delta[l] of the layers below the cuts (backward: the
 * upper segment leaves d loss / d h there, the lower one overwrites it with delta) must then be kept -- EMX_EUNSUPPORTED names the
 * missing one, and a caller whose plan drops rows (NULL entries) falls back to the plan that keeps them.  The launches that merge
 * passes (emx_mlp_forward_cost with a cost, emx_mlp_forward_backward*) answer EMX_EUNSUPPORTED for such a network. */
int emx_mlp_forward(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                    int ld_in, int row_begin, int row_end, float* const* h, void* aop, int aop_rows, emx_stream_t stream);
/* INFERENCE: layers [layer_begin, layer_end) on n_rows rows as ONE streaming launch -- nothing but the input rows and the result
 * touches HBM (csrc/infer.hip).
 *   replaces: Autoencoder.encode / generate / decode   encodermap/autoencoder/autoencoder.py:1110-1195 (model.encoder / model.decoder on
 *             whole trajectories), SequentialModel.encoder / decoder  encodermap/models/models.py:3331-3359
 *   in  [n_rows, width[layer_begin]] (row stride ld_in floats), or -- periodicity_in finite and > 0 -- the RAW angles
 *       [n_rows, width[layer_begin] / 2]: the network input [sin(x 2 pi / p) | cos(x 2 pi / p)] (PeriodicInput, layers.py:204-215) is
 *       formed on chip;
 *   out [n_rows, width[layer_end]] (row stride ld_out), or -- periodicity_out finite and > 0 -- the angles
 *       atan2(y[:, k], y[:, n + k]) p / 2 pi, [n_rows, n = width[layer_end] / 2] (PeriodicOutput, layers.py:1118-1125);
 *   0 or INFINITY for either periodicity: plain rows.  No operand store, no fragment store, no workspace: every workgroup builds the
 *   split-f16 operand pieces of the range in its LDS from `params` (emx_mlp_desc offsets).  Same precision class as emx_mlp_forward,
 *   another order of summation (agreement to float32 rounding, not bit for bit).
 * Widths: every layer at most 128 x 128 (pieces of the whole range resident in LDS) -- or ONE wide end, a first layer of any
 * number of inputs or a last layer of any number of out-features, on the standard shape [wide end, tanh layer of 65..128 units, tanh
 * layer of 65..128 units, other end] (n_neurons [128, 128, L] of parameters.py:611-639 on wide data: the wide layer's pieces are
 * staged through LDS chunk by chunk).
 * EMX_EUNSUPPORTED (nothing has been launched): any other layer wider than 128, or pieces beyond 160 KB of LDS (e.g. three
 * 128 x 128 layers) -- run emx_mlp_forward (+ emx_periodic_in / emx_periodic_out) instead. */
int emx_mlp_infer(const emx_mlp_desc* desc, const float* params, int layer_begin, int layer_end, const float* in, int64_t ld_in,
                  float periodicity_in, int64_t n_rows, float* out, int64_t ld_out, float periodicity_out, emx_stream_t stream);
/* emx_mlp_forward with the auto cost folded into the same launch (a cost kernel of its own is ~5 us of launch and first-touch
 * latency for a few hundred kilobytes): the cost of emx_periodic_cost_fwd_bwd is evaluated on the rows of layer
 * layer_end-1 while they are still on chip.  dY [*, width[layer_end]] receives d cost / d output; ws[g],
 * g < emx_mlp_row_blocks(row_end - row_begin), the row blocks' shares of the cost, already scaled by coef / (B n) (or
 * coef / B for EMX_MEAN_NORM) -- their fixed-order sum (emx_finish_step) is the cost.  cost == NULL: emx_mlp_forward. */
typedef struct {
  const float* target;    /* [*, ldt] */
  int ldt;
  int n;                  /* compared columns */
  int off_sin, off_cos;   /* column windows of the prediction inside the output row (off_cos only for periodic_output) */
  int periodic_output;    /* the output row holds [sin | cos] pairs that are turned into angles first */
  float periodicity;      /* INFINITY: plain differences; < 0 (finite): the periodic output's period is |periodicity| but the
                             differences are plain, not wrapped (reconstruction_loss / "mse" behind PeriodicOutput) */
  int variant;            /* EMX_MEAN_ABS / EMX_MEAN_SQUARE / EMX_MEAN_NORM */
  float coef;
  int B;                  /* global batch size of the normalisation */
  float* dY;
  float* ws;
} emx_auto_cost;
int emx_mlp_row_blocks(int n_rows);
int emx_mlp_forward_cost(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                         int ld_in, int row_begin, int row_end, float* const* h, const emx_auto_cost* cost, void* aop, int aop_rows,
                         emx_stream_t stream);
/* Forward pass, auto cost and the backward-data pass through the upper layers [bwd_begin, layer_end) in ONE launch: the
 * row block never leaves the CU between them (its activations stay in LDS), which saves the second launch's start-up
 * (the weights' first-touch latency, ~4 us) and the HBM round trip of d cost / d output.
 *   h[l] (all forward layers) and delta[l] (bwd_begin <= l < layer_end) are written as for the separate calls;
 *   delta[layer_end - 1] = d cost / d output (the top layer must be linear); dbelow (may be NULL) receives dh below
 *   layer bwd_begin -- exactly what emx_mlp_forward_cost followed by emx_mlp_backward(bwd_begin, layer_end, dtop = dY,
 *   dbelow) produce.  EMX_EUNSUPPORTED when the row block's LDS images do not fit (then use the two calls). */
int emx_mlp_forward_backward(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                             int ld_in, int row_begin, int row_end, float* const* h, const emx_auto_cost* cost, int bwd_begin,
                             float* const* delta, float* dbelow, void* aop, int aop_rows, emx_stream_t stream);
/* Two kernels stand behind that entry point.  A network whose every width is <= 128, with tanh / linear layers, a non-periodic
 * output and the mean_abs or mean_square cost, called on the whole layer range with at most one row block per compute unit, runs as
 * the TRANSPOSED chain (csrc/tchain.h: H^T = W^T X^T, a layer's accumulators are the next layer's operand registers); everything
 * else as the row-block chain kernel.  Same precision class (split-f16 x 3 products, float32 accumulation), different order of
 * summation: results agree to float32 rounding, not bit for bit.  Environment EMX_TCHAIN=0 (read per call) keeps the row-block
 * kernel everywhere, EMX_TCHAIN=1 takes the transposed chain at any number of row blocks.  emx_transposed_chain_count(): launches of
 * the transposed chain by this process so far (diagnostics, like emx_launch_count). */
int64_t emx_transposed_chain_count(void);
/* The same launch with the ALL-PAIRS TILES of the sketch-map cost aboard (emx_sketchmap_tiles on the batch's split rows; replaces the
 * serial order of SequentialModel.train_step, models/models.py:3367-3401, where loss_functions.py:266-296's distance cost is evaluated
 * after the whole forward pass): the tile pass needs the batch rows and the latent only, so its tiles run as extra workgroups of
 * this launch -- Gram products at once, the sigmoid epilogue as soon as the workgroups of the dense layers have PUBLISHED the latent
 * rows of layer bwd_begin - 1 (the encoder's last), i.e. beside the decoder's forward pass, the auto cost and the decoder's backward
 * pass.  Outputs: everything emx_mlp_forward_backward leaves, plus the per-tile partial sums and tile costs in sk_ws exactly as
 * emx_sketchmap_tiles(X, B, D, ldx, h[bwd_begin - 1], 2, INFINITY, ., sig, 0, B, sk_ws, ., split_ws) leaves them (bit for bit).
 *   rows [row_begin, row_end) must be the whole batch [0, B); X / D / ldx describe the batch rows, periodicity / data_span as for
 *   emx_sketchmap_tiles.  Euclidean X (periodicity INFINITY): split_ws = their split planes (emx_sketchmap_split / emx_next_batch),
 *   the Gram tiles of the f16 matrix cores.  Periodic X: the 32 x 32 VALU tiles on the rows themselves (batches of up to 512 rows); zpub: B x 2 floats of scratch for the published latent rows; sync: 4
 *   uint32 counter words (zeroed by the library at first sight, see "counter words"; the launch leaves [0] and [1] at zero; [2] != 0
 *   afterwards: a tile workgroup gave up waiting for the latent -- the results of that launch and of every later one are invalid).
 *   Every dense-layer workgroup and at least one tile workgroup must be resident at once (asked of the runtime for the launch's
 *   instantiation and LDS on the current device): EMX_EUNSUPPORTED otherwise.
 * EMX_EUNSUPPORTED outside what the tile role is built for (latent width 2; Euclidean rows of 32 / 64 / 128 / 256 features -- up to 128
 * with sigmoid exponents outside the (a, b) = (12, 6 | 2, 6) family; periodic batches of up to 512 rows; LDS) -- then use emx_mlp_forward_backward + emx_sketchmap_tiles. */
int emx_mlp_forward_backward_tiles(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                                   int ld_in, int row_begin, int row_end, float* const* h, const emx_auto_cost* cost, int bwd_begin,
                                   float* const* delta, float* dbelow, void* aop, int aop_rows, const float* X, int D, int ldx, float periodicity,
                                   float data_span, const float sig[6], void* sk_ws, size_t sk_ws_bytes, void* split_ws /* NULL for periodic X */,
                                   float* zpub, uint32_t* sync, emx_stream_t stream);
/* emx_mlp_forward_backward_tiles with the backward units BELOW the latent layer in the same launch as well (batches whose tiles ride in
 * the fused launch: the training step's first three launches -- dense layers forward + auto cost + decoder backward, all-pairs tiles,
 * encoder backward -- as ONE): bwd_begin may reach down to layer_begin (0: all of the encoder), latent_layer names the layer whose rows
 * the tile workgroups wait for, and `latent` describes where they leave the partial sums of the all-pairs cost's gradient
 * (emx_sketchmap_describe on sk_ws).  In front of the epilogue that forms dh of the latent layer every dense-layer workgroup waits
 * -- one lane polls -- until the last tile workgroup has reported, collects its rows of that gradient as emx_mlp_backward's `latent`
 * join does (same order of additions) and carries on down the encoder from the images it still holds in LDS.
 *   sync: 8 uint32 counter words, zeroed by the library at first sight ([3] all tile workgroups done, [4] dense-layer workgroups
 *   through: left at zero by the launch; [2] sticky: a wait gave up).
 *   Same results as emx_mlp_forward_backward_tiles (bwd_begin = latent_layer + 1, dbelow) followed by emx_mlp_backward(layers
 *   [layer_begin, latent_layer + 1), dtop = dbelow, latent) to float32 rounding of the latent layer's image (the one launch forms
 *   it in its epilogue, the two in the second launch's staging). */
int emx_mlp_forward_backward_tiles_join(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                                        int ld_in, int row_begin, int row_end, float* const* h, const emx_auto_cost* cost, int bwd_begin,
                                        float* const* delta, float* dbelow /* may be NULL */, void* aop, int aop_rows, const float* X, int D, int ldx,
                                        float periodicity, float data_span, const float sig[6], void* sk_ws, size_t sk_ws_bytes,
                                        void* split_ws /* NULL for periodic X */, float* zpub, uint32_t* sync, int latent_layer,
                                        const emx_latent_grad* latent, emx_stream_t stream);
/* dtop = d loss / d h[layer_end-1] (row stride ld_top).  For l = layer_end-1 .. layer_begin:
 *   delta[l] = (dh_l + (l == layer_add ? dadd : 0)) * act'(h[l]);  dh_{l-1} = delta[l] W_l^T.   dadd may be NULL.
 *   dbelow (may be NULL): [B, width[layer_begin]] receives dh of the layer below the range, so that the decoder and
 *   the encoder can be back-propagated by separate launches (the decoder's overlaps the all-pairs kernel). */
int emx_mlp_backward(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* dtop,
                     int ld_top, const float* dadd, int layer_add, const emx_latent_grad* latent /* may be NULL; joins at layer_add */,
                     float* dbelow, int row_begin, int row_end, float* const* h, float* const* delta, void* aop, int aop_rows,
                     emx_stream_t stream);
/* grad[w_off[l]..] = X_l^T delta[l], grad[b_off[l]..] = column sums of delta[l] over rows [row_begin,row_end), for
 * layers [layer_begin, layer_end), from the fragments the forward and backward launches left in `aop` for these rows
 * (X_0 = the network input, X_l = h[l-1]).  Two launches: per-slice partials (EMX_WGRAD_SLICE_ROWS rows each, more for
 * batches beyond EMX_WGRAD_SLICE_ROWS * EMX_WGRAD_MAX_SLICES rows) into ws, then a fixed-order sum.
 * ws: emx_mlp_wgrad_ws_floats(n_params, n_rows) floats, ZERO-INITIALISED ONCE by the caller (alignment padding between
 * tensors is never written). */
size_t emx_mlp_wgrad_ws_floats(int64_t n_params, int n_rows);
int emx_mlp_wgrad(const emx_mlp_desc* desc, float* params, int64_t n_params, int layer_begin, int layer_end, const void* aop, int aop_rows,
                  int row_begin, int row_end, float* grad /* NULL: leave the slice partials in ws for emx_adam_step_slices */,
                  float* ws, size_t ws_floats, const emx_finish_desc* finish /* may be NULL: emx_finish_step with these
                  arguments, run by one extra workgroup of this launch */, emx_stream_t stream);
/* number of row slices emx_mlp_wgrad leaves in ws for n_rows rows (slice s starts at ws + s * n_params) */
int emx_mlp_wgrad_slices(int n_rows);

/* A training step's last two launches in one, for small batches: emx_mlp_wgrad (all layers, with its `finish` epilogue) and
 * emx_adam_step_slices(..., bookkeeping_done = 1) with `next`, i.e. tape.gradient wrt the kernels and biases, Keras History / counters,
 * Adam(lr, clipvalue).apply_gradients and the next tf.data batch (models.py:3392-3401, autoencoder.py:741-743,845-856) -- weight
 * gradients of all layers, the `finish` epilogue, L2 + clip + Adam with the operand store, the next batch's rows -- in one launch
 * WITHOUT slice partials: a workgroup owns one 32 x 32 block of one kernel over ALL rows
 * (its four waves a quarter of the row blocks each, summed in LDS in a fixed order) and updates those parameters itself; no
 * workspace, no workgroup waits for another.  Deterministic; the gradient's summation order differs from emx_mlp_wgrad's, so
 * results agree with the two launches to float32 rounding of the sums, not bit for bit.
 *   finish: required, with hstep, cursor and counter2 (= the optimizer's step counter, read as `step`); cursor and optimizer
 *     step are advanced by the epilogue workgroup once every other workgroup has read them.
 *   reg_ws: 2 * EMX_REDUCE_WS floats: the regulariser shares of the weights after update t live in half (t & 1) -- the
 *     epilogue of a launch reads the half its predecessor wrote while the launch writes the other.  A `finish` job whose
 *     src == reg_ws sums that half.  At start-up (and whenever the weights are set from outside) fill BOTH halves
 *     (emx_l2_partials on reg_ws and on reg_ws + EMX_REDUCE_WS).
 *   sync: 4 + 1024 uint32 counter words (zeroed by the library at first sight); the launch leaves [1] and [4 ..] zero again, [2] != 0
 *     (sticky): the epilogue gave up waiting -- the count is then left as it is, and this and every later step are invalid.
 *   ws: emx_mlp_wgrad_adam_local_ws_floats(desc, n_rows) floats (0 for small batches: ws may be NULL).  Batches of more than 512
 *     rows split the rows of a block over up to four workgroups (the launch is bound by the bytes a CU can have in flight, so by
 *     the number of CUs at work): each leaves its share in ws, the LAST to arrive adds them in the order of the parts.
 * emx_mlp_wgrad_adam_local_fits: 1 if the launch is made for this network / batch (rows <= 4096, at most 1000 blocks of 32 x 32 over all kernels, a `next`
 * whose rows allow 16-byte lanes and no ADC head), else 0 -- emx_mlp_wgrad_adam_local fails with EMX_EUNSUPPORTED then. */
int emx_mlp_wgrad_adam_local_fits(const emx_mlp_desc* desc, int n_rows, const emx_next_batch* next /* may be NULL */);
size_t emx_mlp_wgrad_adam_local_ws_floats(const emx_mlp_desc* desc, int n_rows);
int emx_mlp_wgrad_adam_local(const emx_mlp_desc* desc, float* params, int64_t n_params, const void* aop, int aop_rows, int row_begin, int row_end,
                             float* ws, size_t ws_floats, const emx_finish_desc* finish, float* m, float* v, const float* l2coef /* may be NULL */, float lr, float beta1,
                             float beta2, float eps, float clip, float* reg_ws, float* mirror, const emx_next_batch* next /* may be NULL */,
                             uint32_t* sync, emx_stream_t stream);

/* ------------------------------------------------------------------------------------------------ *
 * ADC path (AngleDihedralCartesianEncoderMap): back-mapping and per-frame pair distances
 * ------------------------------------------------------------------------------------------------ */
/* K8  column mean over the batch: out[k] = mean_b x[b,k]     MeanAngles / mean bond lengths, layers.py:971,1153-1161 */
int emx_colmean(const float* x, int B, int n, int ldx, float* out, emx_stream_t stream);

/* K6  chain_in_plane   encodermap/encodermap_tf1/backmapping.py:97-119
 *   lengths [A-1] (already the batch mean), angles [B, A-2] (stride lda; lda = 0 broadcasts one row of
 *   batch-mean angles to every frame, MeanAngles layers.py:1153-1161) -> xyz [B, A, 3] (z = 0)
 *   bwd: dxyz [B,A,3] -> dangles [B, A-2] (stride ldd), WRITTEN.  (lengths are data: no gradient) */
int emx_chain_in_plane_fwd(const float* lengths, const float* angles, int lda, int B, int A, float* xyz, emx_stream_t stream);
int emx_chain_in_plane_bwd(const float* lengths, const float* angles, int lda, const float* dxyz, int B, int A,
                           float* dangles, int ldd, emx_stream_t stream);

/* K7  dihedrals_to_cartesian (two-way, split at A/2)   encodermap/encodermap_tf1/backmapping.py:164-214,
 *     encodermap/misc/backmapping.py:179-309,1873-1912 ; rotation_matrix encodermap/encodermap_tf1/misc.py:286-304
 *   dihedrals [B, A-3] (stride ldd; `dihedral_offset` (= pi in BackMapLayer, layers.py:979) is added inside),
 *   plane [B, A, 3] -> out [B, A, 3].   bwd: dout -> ddihedrals [B, A-3] (stride ldg) and dplane [B, A, 3], WRITTEN. */
int emx_dihedrals_to_cartesian_fwd(const float* dihedrals, int ldd, float dihedral_offset, const float* plane,
                                   int B, int A, float* out, emx_stream_t stream);
int emx_dihedrals_to_cartesian_bwd(const float* dihedrals, int ldd, float dihedral_offset, const float* plane,
                                   const float* dout, int B, int A, float* ddihedrals, int ldg, float* dplane,
                                   emx_stream_t stream);

/* K3  per-frame flat pairwise distances of atoms start, start+step, ... (C atoms)   PairwiseDistances, layers.py:1252-1267
 *     with pairwise_dist(flat=True) distances.py:235-253.  xyz [B, A, 3]; out [B, ldo] holds the C(C-1)/2 distances of
 *     the strict upper triangle, row-major, followed by zeros up to ldo (so the block can feed the all-pairs kernel
 *     with a 16-byte aligned row).  bwd: dxyz WRITTEN (accumulate = 0; unselected atoms get 0) or ADDED to. */
int emx_pdist_flat_fwd(const float* xyz, int B, int A, int start, int step, int C, float* out, int ldo, emx_stream_t stream);
int emx_pdist_flat_bwd(const float* xyz, const float* dout, int B, int A, int start, int step, int C, int ldo,
                       float* dxyz, int accumulate, emx_stream_t stream);

/* The forward half of the ADC decoder's tail in ONE launch, a workgroup per frame: PeriodicOutput of the angle and dihedral
 * blocks of the decoder output Y [B, ldy] (columns ang_sin / ang_cos / dih_sin / dih_cos; models.py:2355-2420,
 * layers.py:1118-1125), BackMapLayer (chain_in_plane + dihedrals_to_cartesian with `dihedral_offset`, layers.py:957-986) and
 * PairwiseDistancesLayer (layers.py:1252-1267) of the back-mapped frames -> out_pair, and of the input frames `cart` ->
 * inp_pair (inp_pair NULL: skipped).  Equal, value for value, to emx_periodic_out x2 + emx_chain_in_plane_fwd +
 * emx_dihedrals_to_cartesian_fwd + emx_pdist_flat_fwd x2; every intermediate is written as those calls write it
 * (ang_pred [B, A-2], dih_pred [B, A-3], plane, xyz [B, A, 3], pair blocks [B, ldo]).  ang_sin < 0: the angles are the batch
 * means `mean_angles` [A-2] (MeanAngles, layers.py:1153-1161; ang_pred is not written).  mean_lengths [A-1].
 * split_ws (may be NULL; emx_sketchmap_split_bytes(B, C(C-1)/2) bytes): additionally emx_sketchmap_split of inp_pair, the
 * operand planes of the all-pairs kernel for the cartesian distance cost (its high-d points are the input frames' pair
 * distances, models.py:2448-2475), bit for bit.
 * EMX_EUNSUPPORTED when a frame does not fit the LDS buffers of the fused pass (the single stages remain). */
int emx_adc_frame_fwd(const float* Y, int ldy, int B, float periodicity, int ang_sin, int ang_cos, const float* mean_angles,
                      float* ang_pred, int dih_sin, int dih_cos, float* dih_pred, const float* mean_lengths,
                      float dihedral_offset, int A, float* plane, float* xyz, const float* cart, int start, int step, int C,
                      float* inp_pair, float* out_pair, int ldo, void* split_ws, emx_stream_t stream);

/* ... and the way back in ONE launch, a workgroup per frame: the cartesian cost (models.py:2405-2447: the row-mean cost
 * `variant` with `coef` [x *coef_scale, the soft-start scale] of out_pair against inp_pair, [B, ldo] each with C(C-1)/2 pair
 * distances per row), the backward pass of PairwiseDistancesLayer and of dihedrals_to_cartesian (tape.gradient,
 * models.py:3392-3393).  Equal, value for value, to emx_periodic_cost_fwd_bwd (loss NULL: its prescaled workgroup shares in
 * ws [EMX_REDUCE_WS]) + emx_pdist_flat_bwd + emx_dihedrals_to_cartesian_bwd; d cost / d pair distances and d cost / d xyz
 * stay on the chip.  Written: ws, ddihedrals [B, A-3] (stride ldg), dplane [B, A, 3].
 * angles [B, A-2] (may be NULL; models with backbone angles): additionally emx_chain_in_plane_bwd(mean_lengths, angles, dplane)
 * -> dangles [B, A-2] (stride ldd) behind the back-mapping's backward pass, in the same launch.
 * jobs (n_jobs 0 .. EMX_MAX_COST_JOBS, may be NULL): additionally emx_periodic_cost_batch(jobs, n_jobs, B, 0, B) behind it, in
 * the same launch -- the angle-like costs of the decoder output (models.py:2260-2521); a job's dpred_add may be `ddihedrals`
 * or `dangles` (the gradient of the cartesian path joins the cost's, written by this launch).  Jobs of 128 .. 512 columns, not
 * EMX_MEAN_NORM (else EMX_EUNSUPPORTED: call without jobs and emx_periodic_cost_batch).
 * EMX_EUNSUPPORTED outside the shapes the pass is built for (B <= EMX_REDUCE_WS / 4 frames, more than 1536 pair distances
 * per frame, not EMX_MEAN_NORM, a frame that fits LDS): the single stages remain. */
int emx_adc_frame_bwd(const float* out_pair, const float* inp_pair, int ldo, int variant, float coef, const float* coef_scale,
                      float* ws, const float* xyz, const float* dihedrals, float dihedral_offset, const float* plane, int B, int A,
                      int start, int step, int C, float* ddihedrals, int ldg, float* dplane, const float* mean_lengths,
                      const float* angles, float* dangles, int ldd, const emx_cost_job* jobs, int n_jobs, emx_stream_t stream);

/* emx_adc_frame_fwd and emx_adc_frame_bwd as ONE launch, a workgroup per frame (nothing between the two depends on another frame):
 * the arguments of the first, then what the second adds -- the cartesian cost (variant, coef, coef_scale, ws), its outputs
 * (ddihedrals / ldg, dplane, dangles / ldd) and the cost jobs; its out_pair / inp_pair / xyz / dihedrals / plane / angles are the
 * forward half's out_pair / inp_pair (required) / xyz / dih_pred / plane / ang_pred (when the angles come from Y).  Same values, bit for
 * bit, as the two launches; EMX_EUNSUPPORTED where either of them is. */
int emx_adc_frame_fwd_bwd(const float* Y, int ldy, int B, float periodicity, int ang_sin, int ang_cos, const float* mean_angles, float* ang_pred,
                          int dih_sin, int dih_cos, float* dih_pred, const float* mean_lengths, float dihedral_offset, int A, float* plane,
                          float* xyz, const float* cart, int start, int step, int C, float* inp_pair, float* out_pair, int ldo, void* split_ws,
                          int variant, float coef, const float* coef_scale, float* ws, float* ddihedrals, int ldg, float* dplane, float* dangles,
                          int ldd, const emx_cost_job* jobs, int n_jobs, emx_stream_t stream);

/* Training metrics of the ADC model, evaluated on the back-mapped frames [B, A, 3]:
 *   emx_clash_count  counts[f] = number of atom pairs i < j closer than clash_distance -- ADCClashMetric.update_state,
 *                    count_nonzero(pairwise_dist(y_pred, flat=True) < d, axis=1)      encodermap/callbacks/metrics.py:505-522
 *   emx_kabsch_rmsd  rmsd[f] of frame f of a against frame f of b after the optimal rotation, weights tile(weights3, A/3) --
 *                    kabsch_weighted / rmsd / ADCRMSDMetric (metrics.py:71-153, 229-243, 570-581), including the
 *                    reference's `translate` step (every atom shifted by the mean of its own three coordinates). */
int emx_clash_count(const float* xyz, int B, int A, float clash_distance, int64_t* counts, emx_stream_t stream);
int emx_kabsch_rmsd(const float* a, const float* b, int B, int A, const float weights3[3], int translate, float* rmsd, emx_stream_t stream);
/* per-step record inside a captured step (Keras logs every metric's result() each epoch = step):
 * clash_hist[*hstep] = mean(counts) truncated like tf.reduce_mean of int64 (metrics.py:521), rmsd_hist[*hstep] = mean(rmsd);
 * either metric may be NULL; call before emx_finish_step advances *hstep. */
int emx_metric_tick(const int64_t* counts, const float* rmsd, int B, float* clash_hist, float* rmsd_hist, int64_t capacity,
                    const int64_t* hstep, emx_stream_t stream);

/* Featurisation of a backbone chain: xyz [N, A, 3] -> bond lengths [N, A-1], bond angles
 * [N, A-2] (rad) and dihedrals [N, A-3] (rad, mdtraj convention) of consecutive atoms -- the ADC model's inputs, which the
 * reference obtains through mdtraj's C kernels (encodermap/loading/features.py:150-156,1133,1394; formulas restated at
 * encodermap/misc/backmapping.py:328-352, encodermap_tf1/moldata.py:270-271).  Any output may be NULL. */
int emx_backbone_features(const float* xyz, int64_t N, int A, float* distances, float* angles, float* dihedrals,
                          emx_stream_t stream);

/* All-atom back-mapping: every frame starts as a copy of xyz0 (frame_stride 0: one template
 * frame for all, as encodermap/misc/backmapping.py:1607-1613 joins copies of frame 0; otherwise frame f starts from
 * xyz0 + f * frame_stride) and has its torsions set one after the other: for j = 0 .. J-1 the atoms on the far side of the
 * bond quads[j][1] - quads[j][2] are rotated about that bond by targets[f][j] minus the current dihedral of quads[j]
 * (rad; backmapping.py:1668-1690 central dihedrals, :1722-1745 side-chain dihedrals -- the caller lists them after the
 * central ones; numba variant `parallel_rotation_application` :383-405).
 *   quads    int32 [J, 4]                     atom indices of each torsion
 *   far_mask uint32 [J, W, 64], W = ceil(A / 2048) (W * 64 = emx_allatom_mask_words(A)): bit k of far_mask[j][w][l] is set
 *            iff atom (32 w + k) * 64 + l lies on the far side of torsion j (the side of quads[j][2]; the reference finds it
 *            by cutting the bond graph, misc/rotate.py:409-511)
 *   targets  float [n_frames, ldt >= J]       requested dihedrals, rad
 *   out      float [n_frames, A, 3]
 * A is limited to 5461 atoms (the frame lives in LDS). */
int emx_allatom_mask_words(int A);
int emx_allatom_backmap(const float* xyz0, int64_t frame_stride, int64_t n_frames, int A, const int32_t* quads, const uint32_t* far_mask,
                        int J, const float* targets, int64_t ldt, float* out, emx_stream_t stream);

/* soft start of the cartesian cost (callbacks/callbacks.py:584-606) evaluated on the device from the step counter:
 * *out = 0 (t < a), c (t - a)/(b - a) (a <= t <= b), c (t > b); a < 0 -> constant c.   t = *step */
int emx_soft_start_scale(const int64_t* step, float soft_a, float soft_b, float scale_c, float* out, emx_stream_t stream);

/* The head of an ADC training step in ONE launch (gen_functional_model's inputs, models.py:385-1060; tf.data batching,
 * autoencoder.py:1613-1644; IncreaseCartesianCost, callbacks.py:584-606): rows idx[*cursor * B + b] (cursor NULL: idx[b]) of
 *   feat_data [N, F] -> feat [B, F]   (angles | dihedrals | side dihedrals | distances)      emx_gather_rows
 *   cart_data [N, C3] -> cart [B, C3] (flattened Cartesians)                                  emx_gather_rows
 *   for each of the n_parts <= 4 angular blocks (part_off, part_n) of the feature row: sin / cos into
 *   xin[b, part_col .. part_col + 2 part_n)                                                    emx_periodic_in
 *   *cart_scale = soft-start scale of step *step (cart_scale NULL: skipped)                   emx_soft_start_scale
 *   feat_mean [F] = column means of the gathered feature rows (NULL: skipped)                  emx_colmean
 *     (mean bond lengths of BackMapLayer, layers.py:971; MeanAngles, layers.py:1153-1161)
 * -- the same numbers as those entry points, one launch instead of five. */
/* emx_adc_batch_head's description as a device record (emx_adc_head_bytes() bytes at `record`, written with a blocking copy) for
 * emx_next_batch::adc_head: the optimizer launch of step t then does the head of step t + 1 on workgroups of its own (the rows
 * idx[(*cursor + 1) * B + b], skipped behind the index block), as it does the row gather of the EncoderMap step. */
size_t emx_adc_head_bytes(void);
int emx_adc_head_describe(const float* feat_data, const float* cart_data, int64_t N, int F, int C3, int B, float* feat, float* cart,
                          const int* part_off, const int* part_n, const int* part_col, int n_parts, float periodicity, float* xin,
                          int ld_xin, const int64_t* step, float soft_a, float soft_b, float scale_c, float* cart_scale,
                          float* feat_mean, int n_batches, void* record);
int emx_adc_batch_head(const float* feat_data, const float* cart_data, int64_t N, int F, int C3, const int64_t* idx, const int64_t* cursor,
                       int B, float* feat, float* cart, const int* part_off, const int* part_n, const int* part_col, int n_parts,
                       float periodicity, float* xin, int ld_xin, const int64_t* step, float soft_a, float soft_b, float scale_c,
                       float* cart_scale, float* feat_mean, emx_stream_t stream);

/* One-shot all-reduce of a small float buffer between the ranks of one node over peer-mapped memory (csrc/p2p.hip): the data-parallel
 * step's [gradient | cost terms] buffer (one collective per step; the reference itself is single-process,
 * autoencoder.py:741-748).  Every rank publishes the buffer in uncached device memory its peers have mapped through the HIP IPC
 * interface, signals, reads all peers' copies over xGMI and adds them IN RANK ORDER -- bit-identical results on every rank.  One
 * launch per call, no host synchronisation, capturable in a hipGraph.  Protocol of use, the same on every rank:
 *   emx_p2p_create   allocates this rank's region for buffers of up to n_floats floats, `chunks` workgroups per launch
 *                    (1 .. EMX_P2P_MAX_CHUNKS; the same on every rank), and writes its IPC handle (emx_p2p_handle_bytes() bytes);
 *   (the caller exchanges the handles: any transport -- the engine uses torch.distributed's object all-gather)
 *   emx_p2p_connect  with the world x handle_bytes handles in rank order (the own entry is ignored);
 *   emx_p2p_allreduce(ctx, buf, n, scale, stream): buf[i] <- scale * sum over ranks of buf[i], in place; every rank must issue the
 *                    same sequence of calls (n <= n_floats, 16-byte aligned buf);
 *   emx_p2p_status   *gave_up != 0 (sticky): a wait for a peer ended after ~20 s of polling -- results from then on are invalid;
 *   emx_p2p_destroy.
 * Ranks may share a device (separate processes); ranks of other nodes are not reachable this way. */
#define EMX_P2P_MAX_CHUNKS 64
int emx_p2p_handle_bytes(void);
int emx_p2p_create(int world, int rank, int64_t n_floats, int chunks, void** ctx_out, void* handle_out);
int emx_p2p_connect(void* ctx, const void* handles);
int emx_p2p_allreduce(void* ctx, float* buf, int64_t n, float scale, emx_stream_t stream);
int emx_p2p_status(void* ctx, int* gave_up);
int emx_p2p_destroy(void* ctx);

#ifdef __cplusplus
}
#endif
#endif /* EMX_H */
