/*
 * wfsparse.h -- C ABI of libwfsparse.so: the MI355X (gfx950) sparse-convolution hot path of
 * WaveformML's LitPSD training step.
 *
 * This is the drop-in boundary.  Every entry point replaces one operator of the third-party
 * package the reference calls for this path, spconv~=1.2.1 (reference requirements.txt:15), i.e.
 * the torch custom ops behind its Python surface:
 *
 *   torch.ops.spconv.get_indice_pairs      <- spconv.ops.get_indice_pairs, called from
 *       SparseConvolution.forward for every layer the reference constructs at
 *       src/models/SPConvBlocks.py:75,134,191,249,298,335,370,498,502,803,809 and by config
 *       strings "spconv.SubMConv3d" etc. (src/utils/ModelValidation.py:24-31)
 *   torch.ops.spconv.indice_conv           <- spconv.functional.indice_conv / indice_subm_conv /
 *   torch.ops.spconv.indice_conv_backward     indice_inverse_conv (same call sites; inverse conv
 *                                             at src/models/SPConvBlocks.py:804,810)
 *   SparseConvTensor.dense()               <- spconv.ToDense (src/models/SPConvBlocks.py:81,515),
 *                                             src/engineering/LitBase.py:138-146
 *
 * Conventions
 *   - plain C types only: raw DEVICE pointers (HBM), sizes, and an opaque hipStream_t passed as
 *     void*.  No torch types.  The library owns no memory and keeps no global state besides a
 *     thread-local error string and an opt-in event-timing table; all outputs and workspaces
 *     are caller-allocated (two-phase "plan" -> "run" where a size is data dependent).
 *   - every call is asynchronous on `stream` unless documented otherwise.
 *   - return value: 0 = WFS_OK, otherwise one of WFS_E*; wfs_last_error() gives the text.
 *     The Python shim maps WFS_EINVAL -> AssertionError / RuntimeError as spconv raises them
 *     (SURVEY.md 8b "Error conventions").
 *   - index tensors are int32, batch-first [N, ndim+1] exactly as spconv's (the reference
 *     permutes its (x,y[,t],evt) columns to batch-first at src/models/SPConvNet.py:47-52,64).
 *   - feature dtype codes: WFS_F32 (fp32 storage, fp32 accumulate), WFS_BF16 and WFS_F16 (16-bit
 *     storage, fp32 accumulate; the reference's `half_precision` rows are fp16,
 *     src/datasets/HDF5Dataset.py:227).  Filters [K, Cin, Cout] are fp32 in every case (master
 *     weights, rounded to the storage type while they are staged for the matrix cores).
 *   - device-side row counts: every row-dimensioned call takes its row count BY VALUE (array
 *     strides, grid size, = the capacity) and an optional `const int64_t *..._dev` that, when not
 *     NULL, holds the number of VALID rows (<= the capacity) in device memory.  Rows beyond it are
 *     neither read nor written.  With device counts no call needs the host to know a data-dependent
 *     size, so a whole training step can be captured into one HIP graph and replayed.
 */
#ifndef WFSPARSE_H
#define WFSPARSE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WFS_OK 0
#define WFS_EINVAL 1      /* bad argument / unsupported configuration                    */
#define WFS_EOVERFLOW 2   /* batch * prod(out_shape) >= 2^31 (spconv asserts the same)    */
#define WFS_EHIP 3        /* a HIP runtime call failed                                    */
#define WFS_EWORKSPACE 4  /* workspace too small                                          */

#define WFS_F32 0
#define WFS_BF16 1
#define WFS_F16 2

#define WFS_MAX_DIM 4

/* library / device ------------------------------------------------------------------------
 * WFS_ABI_VERSION changes whenever a struct layout, an exported signature or the meaning of an argument does (3: wfs_geometry grew
 * `transposed` / `output_padding` in round 2, the event-local build joined in round 3; 4: the wide-layer entry points; 5: the failure flags of wfs_rulebook_emit and
 * wfs_event_rulebook_subm are sticky -- set, never cleared, by the library; 6: round 4 -- wfs_gather_conv / wfs_gather_dw take
 * `packed_kl`, the event-local build of regular convolutions and the packed by-input table joined).  A binding
 * compiled against another version must refuse the library: waveformml_amd/_lib.py does. */
#define WFS_ABI_VERSION 6
int wfs_abi_version(void);
const char *wfs_last_error(void);

/* Geometry of one sparse convolution, host side (all arrays have ndim entries).
 * For SubM the front door forces stride = 1, padding = ksize/2, out_shape = spatial
 * (SURVEY.md A.2); call wfs_geometry_init to apply those rules and validate.              */
typedef struct wfs_geometry {
    int32_t ndim;
    int32_t batch_size;
    int32_t subm;
    int32_t K;                          /* prod(ksize), filled by wfs_geometry_init          */
    int32_t spatial[WFS_MAX_DIM];       /* input spatial shape                               */
    int32_t out_shape[WFS_MAX_DIM];     /* filled by wfs_geometry_init                       */
    int32_t ksize[WFS_MAX_DIM];
    int32_t stride[WFS_MAX_DIM];
    int32_t padding[WFS_MAX_DIM];
    int32_t dilation[WFS_MAX_DIM];
    int32_t transposed;                 /* != 0: spconv's SparseConvTranspose geometry (subm must be 0)   */
    int32_t output_padding[WFS_MAX_DIM];/* transposed only                                                  */
} wfs_geometry;

/* Validates and completes a geometry (replaces the Python front door of
 * spconv.ops.get_indice_pairs, A.2).  WFS_EINVAL if a dim has stride>1 and dilation>1,
 * WFS_EOVERFLOW if batch*prod(out_shape) >= 2^31.
 * transposed != 0 (spconv.SparseConvTranspose{2,3}d -- named by the reference at src/utils/ModelValidation.py:30-31):
 * out_shape = (i - 1) s - 2 p + k + output_padding (spconv's get_deconv_output_size: dilation does not enter), input
 * (x, offset o) reaches output x s - p + o d, outputs numbered first-seen with the offsets of a row visited from the
 * LAST to the first (getValidOutPosTranspose walks from the upper corner down).  PARITY UNPINNED on that order: the
 * spconv source is not available here and the reference never constructs one; values are pinned by dense
 * torch conv_transpose (tests).                                                                                       */
int wfs_geometry_init(wfs_geometry *g);

/* rulebook ---------------------------------------------------------------------------------
 * Replaces torch.ops.spconv.get_indice_pairs.  Besides spconv's own encoding
 * (indice_pairs int32 [2,K,N] padded with -1, indice_pair_num int32 [K], both bit-identical
 * to the CPU algorithm of A.3, including first-seen output numbering) the build produces the
 * gather tables the compute kernels consume:
 *     nbr_out int32 [K, N] : for input row j and kernel offset k, the output row, or -1
 *     nbr_in  int32 [K, M] : for output row i and kernel offset k, the input row, or -1
 * (SubM with odd kernels and dilation 1: nbr_in[k] == nbr_out[K-1-k], so nbr_in may be NULL.)
 *
 * Phase 1  wfs_rulebook_plan : site table build (hash or direct grid), candidate lookup,
 *          first-seen numbering.  SubM: nbr_out is final.  Regular conv: nbr_out holds table
 *          slots until phase 2.  Synchronises `stream` ONCE to return, on the host,
 *          host_info = {M, input_has_duplicate_coordinates (SubM only)}.  For SubM host_info may
 *          be NULL: no read-back, no synchronisation -- the caller then vouches that every index
 *          row is in range and that sites are distinct (spconv itself checks neither).
 * Phase 2  wfs_rulebook_emit : regular conv: writes out_indices [M, ndim+1], finalises nbr_out,
 *          fills nbr_in [K, M] if given.  SubM: fills nbr_in if given (needed only for even
 *          kernels / dilation).  Both: spconv's indice_pairs / indice_pair_num if given
 *          (indice_pair_num alone is allowed; pass NULL for both to skip the compaction).
 *
 * `workspace` must hold wfs_rulebook_workspace_bytes(g, N) bytes and stay untouched between
 * the two phases.  WFS_EINVAL if an index row lies outside batch_size / spatial.
 *
 * Device-count mode (no host synchronisation at all): host_info = NULL, n_dev = valid input rows,
 * m_dev = where the plan writes min(M, M_cap) for the consumers of the outputs, M_cap = the caller's
 * output capacity (rows of out_indices / nbr_in handed to emit as M); emit sets *overflow_dev = 1
 * if the true M exceeded it (the step's results are then invalid and must be redone with room) and
 * leaves it alone otherwise: the flag is STICKY, the caller clears it before the first build and
 * after reading it (a captured step replays many builds between two reads). */
size_t wfs_rulebook_workspace_bytes(const wfs_geometry *g, int64_t N);

int wfs_rulebook_plan(const wfs_geometry *g, const int32_t *indices, int64_t N,
                      int32_t *nbr_out, void *workspace, size_t workspace_bytes,
                      int64_t host_info[2], const int64_t *n_dev, int64_t *m_dev, int64_t M_cap,
                      void *stream);

int wfs_rulebook_emit(const wfs_geometry *g, const int32_t *indices, int64_t N, int64_t M,
                      int32_t *nbr_out, int32_t *out_indices, int32_t *nbr_in,
                      int32_t *indice_pairs, int32_t *indice_pair_num,
                      void *workspace, size_t workspace_bytes, const int64_t *n_dev,
                      int32_t *overflow_dev, void *stream);

/* Duplicate-coordinate / range check of an index set whose uniqueness is unknown before a
 * REGULAR conv (SubM learns it for free in its plan; a regular conv's output is unique by
 * construction).  g_subm: a SubM geometry over the set's own spatial shape; workspace as for
 * that geometry.  Synchronises.  host_info = {N, has_duplicates}.                           */
int wfs_indices_check(const wfs_geometry *g_subm, const int32_t *indices, int64_t N,
                      void *workspace, size_t workspace_bytes, int64_t host_info[2], void *stream);

/* gather - GEMM - (no) scatter --------------------------------------------------------------
 * Replaces torch.ops.spconv.indice_conv (forward) and the dX half of indice_conv_backward.
 * Output-stationary: every output row r gathers its <= K source rows through
 * table [K, R] (entry -1 = no neighbour) and contracts them with the per-offset filter:
 *     Y[r, :] = bias + sum_k  X[table[kmap[k], r], :] . W[k]            (transpose_w == 0)
 *     Y[r, :] =        sum_k  X[table[kmap[k], r], :] . W[k]^T          (transpose_w == 1)
 * W is fp32 [K, Cw_in, Cw_out]; with transpose_w the contraction runs over Cw_out.
 * kmap (host array of K ints, may be NULL = identity) lets SubM reuse nbr_out as nbr_in.
 * identity_k >= 0 names the offset whose source row is r itself (SubM centre: spconv computes
 * it as a plain X.W[k*] with k* = argmax indice_pair_num, A.4); -1 = none.  table may be NULL
 * only for K == 1 && identity_k == 0.
 * No atomics: each output row is written exactly once, results are run-to-run reproducible.
 * Packed tables (round 4).  Where a regular conv's kernel is no longer than its stride along the LAST dimension
 * (the PSD nets' SparseConv3d k 3, stride (1,1,4): reference src/models/SPConvBlocks.py:498 through config strings) an
 * input row reaches at most ONE output cell per leading kernel offset q, so the by-input table of
 * wfs_event_rulebook_conv is [K / kl, R] instead of [K, R]: entry e >= 0 means "row e >> 3, at kernel offset
 * k = q * kl + (e & 7)", -1 = none (9 instead of 27 table rows at that geometry, 54 % instead of 18 % of them used).
 * packed_kl = kl hands such a table to wfs_gather_conv (transpose_w products, i.e. dX) and wfs_gather_dw; 0 = the
 * dense form.  wfs_gather_packed_ok(kl, K, Ca, Cb, dtype, which) tells whether a product takes it (which = 1: dX,
 * 2: forward, 3: dW); wfs_unpack_table expands a packed table to the dense one for every other consumer.
 * Arithmetic of fp32 rows (WFS_F32), 32 -> 32 channels (round 4): every fp32 number -- rows and filters -- is cut into
 * three bf16 pieces whose sum is the number exactly, and the six leading piece products are summed in fp32 on
 * v_mfma_f32_16x16x32_bf16 / _32x32x16_bf16 (csrc/conv_mfma.hip k_gconv16_split, k_gdw32_split); what is left out is
 * below 2^-24 of the leading product, the size of one fp32 rounding -- the same 1e-5 bar as the fp32 instructions
 * (v_mfma_f32_16x16x4_f32, taken with the environment's WFS_SPLIT_BF16=0) at 0.6x their time.  Non-finite inputs give
 * NaN (Inf - Inf inside the cut) where an fp32 product would give +-Inf. */
int wfs_gather_conv(const int32_t *table, const int32_t *kmap_host, int32_t K, int32_t identity_k,
                    int64_t R, const void *X, int64_t X_rows, int32_t Cx, const float *W,
                    int32_t Cw_in, int32_t Cw_out, int32_t transpose_w, const float *bias, void *Y,
                    int32_t dtype, const int64_t *r_dev, int32_t packed_kl, void *stream);
int wfs_gather_packed_ok(int32_t packed_kl, int32_t K, int32_t Ca, int32_t Cb, int32_t dtype, int32_t which);
int wfs_unpack_table(const int32_t *packed, int32_t K, int32_t packed_kl, int64_t R, const int64_t *r_dev,
                     int32_t *dense, void *stream);

/* Wide layers as dense matrix-core products (round 3; csrc/wide.hip).
 * Same contract as wfs_gather_conv -- replaces torch.ops.spconv.indice_conv / the dX half of
 * indice_conv_backward (spconv 1.2.1 ops.py; the reference reaches them from src/models/SPConvBlocks.py:450-516
 * with 1697 / 1021 / 345 channels, BASELINE configs[4]) and, with K == 1 && table == NULL, the torch.mm of a
 * 1 x 1 SparseConv2d (spconv conv.py: `features = torch.mm(input.features, weight.view(in, out))`) -- when a side
 * of the filter has >= 128 channels: the layer runs as ONE dense product over whichever side (X_rows source rows,
 * R destination rows) is shorter, on v_mfma_f32_32x32x16_{bf16,f16} for 16-bit rows (filters rounded to the row
 * type) and on v_mfma_f32_32x32x2_f32 for fp32 rows (an exact fp32 fma chain: the 1e-5 path), fp32 accumulate,
 * ordered fp32 sum over the kernel offsets, every output row written once (no atomics).
 * wfs_wide_conv_ok: does this path take the shape?  The workspace holds the filters in the row type, the padded /
 * gathered rows and the per-offset products; 16-byte aligned. */
int wfs_wide_conv_ok(int32_t K, int64_t R, int64_t X_rows, int32_t Cx, int32_t Cy, int32_t dtype);
/* A/B switch for benchmarks (tools/microbench_generic.py): 0 sends every layer back to the 32 x 32-tile kernels
 * (wfs_gather_conv / the narrow arm of wfs_gather_dw); a value >= 8 turns the path on for layers with at least that
 * many channels on a side.  Returns the previous threshold (0 = was off). */
int wfs_wide_enable(int32_t on);
size_t wfs_wide_conv_workspace_bytes(int32_t K, int64_t R, int64_t X_rows, int32_t Cx, int32_t Cy,
                                     int32_t has_table, int32_t dtype);
/* The filters as the products read them: [K][Cw_in][Cw_out rounded up to whole 16-byte pieces] in the row type,
 * zero padded.  A caller that runs several products on the same filters (the forward pass and dX of one layer)
 * converts them once (wfs_wide_filters) and hands the copy to wfs_wide_gather_conv as Wp; with Wp == NULL every
 * call converts W into its workspace (fp32 filters whose rows are whole pieces are read in place). */
size_t wfs_wide_filters_bytes(int32_t K, int32_t Cw_in, int32_t Cw_out, int32_t dtype);
int wfs_wide_filters(const float *W, int32_t K, int32_t Cw_in, int32_t Cw_out, int32_t dtype, void *Wp,
                     void *stream);
int wfs_wide_gather_conv(const int32_t *table, const int32_t *kmap_host, int32_t K, int32_t identity_k,
                         int64_t R, const void *X, int64_t X_rows, int32_t Cx, const float *W, const void *Wp,
                         int32_t Cw_in, int32_t Cw_out, int32_t transpose_w, const float *bias, void *Y,
                         int32_t dtype, const int64_t *r_dev, void *workspace, size_t workspace_bytes,
                         void *stream);

/* Dense head layer on the matrix cores (round 3; csrc/wide.hip).
 * Replaces torch.nn.functional.linear for the reference's dense head when it is wide (the hybrid net's
 * Linear(24150, 269), src/models/SPConvNet.py:40-52 through LinearBlock): y = x W^T + b with x [B, I] in the row
 * type, fp32 W [O, I] (nn.Linear's layout; rounded to the row type for a 16-bit product), fp32 accumulate, fp32 y.
 * The backward takes fp32 dY (rounded to the row type as an operand) and returns dX in the row type, dW / db in
 * fp32; any of the three may be NULL.  Contractions with few output tiles are cut into parts and summed in a fixed
 * order.  wfs_wide_linear_ok: I >= 256, O >= 9 (narrower heads: wfs_head_fwd). */
int wfs_wide_linear_ok(int64_t B, int32_t I, int32_t O, int32_t dtype);
size_t wfs_wide_linear_workspace_bytes(int64_t B, int32_t I, int32_t O, int32_t dtype);
int wfs_wide_linear_fwd(const void *X, int64_t B, int32_t I, const float *W, const float *bias, int32_t O, float *Y,
                        int32_t dtype, void *workspace, size_t workspace_bytes, void *stream);
int wfs_wide_linear_bwd(const void *X, const float *dY, int64_t B, int32_t I, const float *W, int32_t O, void *dX,
                        float *dW, float *db, int32_t dtype, void *workspace, size_t workspace_bytes, void *stream);

/* Event-local rulebook build (round 3; csrc/evrulebook.hip).
 * A sparse convolution never crosses events (the rulebook key includes the batch index) and the
 * reference's collate_fn concatenates the items of a batch in order (src/engineering/PSDDataModule.py:10-20), so the
 * rows of one event are one contiguous range of the index set.
 * wfs_event_offsets: indices int32 [N, ndim + 1] batch-first (n_dev as everywhere) -> offsets int32
 *   [wfs_event_offsets_ints(batch_size)] = { first row of event 0 .. batch_size - 1, number of valid rows,
 *   WFS_EVENT_FLAG_WORDS flag words }; a flag word != 0 <=> the batch column is NOT non-decreasing / in range.  Verified
 *   on the device by every launch; every word is written by every launch; nothing is read back.
 * wfs_event_rulebook_subm: torch.ops.spconv.get_indice_pairs(subm=True) for an index set grouped by event, ONE
 *   WORKGROUP PAIR PER EVENT with the event's site table in LDS (cell -> sample array, direct addressing) -- no site
 *   grid over the batch in HBM, no clearing launch, no global atomics: HBM traffic = the coordinates in, the table out.
 *   nbr_out is bit-identical to wfs_rulebook_plan's.  `events` = wfs_event_offsets of `indices`.
 *   flags: int32 [wfs_event_rulebook_flag_ints(batch)] = 3 x blocks words, STICKY: a launch sets words and never
 *   clears one -- the caller zeroes them before the first build and after reading them (a captured step replays many
 *   builds between two reads).  Any word != 0 in [0, blocks): the index set is not grouped by event or an event exceeds the LDS tables (64 KiB of
 *   sample arrays: 128 active cells at 256 samples) -- nbr_out is then incomplete and the caller takes wfs_rulebook_plan instead;
 *   in [blocks, 2 blocks): duplicate coordinates (same remedy: "the last row wins" is resolved by the chip-wide build);
 *   in [2 blocks, 3 blocks): an index outside the spatial shape.
 *   slots (may be NULL): 64-byte records per row, 32 uint16 slots = 1 + neighbour's row within the event, 0 = none (the
 *   operand form of the event-local conv experiment, tools/exp/event_local/).                                       */
#define WFS_EVENT_FLAG_WORDS 64
size_t wfs_event_offsets_ints(int32_t batch_size);
int wfs_event_offsets(const int32_t *indices, int64_t N, int32_t ndim, int32_t batch_size, const int64_t *n_dev,
                      int32_t *offsets, void *stream);
int wfs_event_rulebook_ok(const wfs_geometry *g);
size_t wfs_event_rulebook_flag_ints(int32_t batch_size);
int wfs_event_rulebook_subm(const wfs_geometry *g, const int32_t *indices, int64_t N, const int64_t *n_dev,
                            const int32_t *events, int32_t *nbr_out, void *slots, int32_t *flags, void *stream);

/* Event-local build of REGULAR (strided) convolutions, ONE launch (round 4; csrc/evconv.hip).
 * torch.ops.spconv.get_indice_pairs(subm=False) -- the reference's SparseConv3d / SparseConv2d layers,
 * src/models/SPConvBlocks.py:75,498 -- for an index set grouped by event, in device-count mode: one workgroup per event
 * keeps the event's OUTPUT grid in LDS (wfs_event_rulebook_conv_ok: <= 16384 cells per event, K <= 32, leading kernel
 * dims <= 4, not transposed), takes first-seen tickets with LDS atomics, numbers the event's sites by a block scan over
 * its rows and learns the ids of the events in front from per-event counts published in `state` (a single pass: no
 * site grid in HBM, no clearing launch, no global read-modify-write, every table written once).  Results are
 * bit-identical to wfs_rulebook_plan + wfs_rulebook_emit (oracle/spconv_ref.c:208-276).
 *   events_in   wfs_event_offsets table of `indices` (or the events_out of the conv that produced them)
 *   M_cap       rows of out_indices / stride of nbr_in; *m_dev = min(M, M_cap); *overflow_dev = 1 (sticky) if M > M_cap
 *   events_out  int32 [wfs_event_offsets_ints(batch)]: the same table for the OUTPUT rows (first-seen numbering of an
 *               event-grouped input is event-grouped)
 *   nbr_out     by-input table: packed_kl == 0: [K, N]; packed_kl == wfs_event_rulebook_conv_packed_kl(g) > 0:
 *               [K / kl, N] packed (see wfs_gather_conv)
 *   nbr_in      by-output table [K, M_cap] (may be NULL)
 *   cell_row    (may be NULL) int32 [batch * out_volume]: output row of every cell or -1 -- what wfs_to_dense_mapped
 *               takes as both `ticket` and `slot_id`
 *   flags       int32 [3], STICKY (set, never cleared): [0] the index set is not grouped by event, or a workgroup
 *               gave up waiting for a count (then *m_dev = 0 / tables invalid: take wfs_rulebook_plan), [2] an index
 *               outside the spatial shape (that row has no outputs)
 *   state       wfs_event_rulebook_conv_state_bytes(batch) bytes, 8-byte aligned, ZEROED once by the caller and then
 *               owned by this layer's builds (launch epoch + per-event counts); two builds must not run concurrently
 *               on one state. */
int wfs_event_rulebook_conv_ok(const wfs_geometry *g);
int wfs_event_rulebook_conv_packed_kl(const wfs_geometry *g);
size_t wfs_event_rulebook_conv_state_bytes(int32_t batch_size);
int wfs_event_rulebook_conv(const wfs_geometry *g, const int32_t *indices, int64_t N, const int64_t *n_dev,
                            const int32_t *events_in, int64_t M_cap, int32_t *out_indices, int64_t *m_dev,
                            int32_t *events_out, int32_t *nbr_out, int32_t packed_kl, int32_t *nbr_in,
                            int32_t *cell_row, int32_t *overflow_dev, int32_t *flags, void *state, void *stream);

/* Replaces the dW half of torch.ops.spconv.indice_conv_backward:
 *     dW[k, a, b] = sum_r  S[r, a] * G[table[k, r], b]          (swap == 0)
 *     dW[k, b, a] = sum_r  S[r, a] * G[table[k, r], b]          (swap == 1)
 * S = the stationary rows [R, Cs], G = the gathered rows [*, Cg]; table column kmap[k] serves
 * offset k (kmap_host NULL = identity; the SubM mirror is accepted for Cs = 32, Cg = 2, which lets
 * the first layer keep its wide dY rows stationary).  Deterministic two-stage reduction through
 * `workspace` (wfs_gather_dw_workspace_bytes).                                                 */
size_t wfs_gather_dw_workspace_bytes(int32_t K, int64_t R, int32_t Cs, int32_t Cg);

/* A pending second stage of wfs_gather_dw: dW = sum over `nslabs` partial results in `part` (the workspace).
 * With `defer` given, wfs_gather_dw runs only its first stage when the shape has a two-stage kernel and describes
 * the rest here (nslabs > 0; dW is NOT written yet, the workspace must stay alive); the caller later reduces the jobs
 * of a whole backward pass in ONE launch with wfs_dw_reduce_jobs -- at the PSD batch sizes every launch costs more
 * than the few hundred KB it reduces.  nslabs == 0 on return: nothing pending, dW is final. */
typedef struct wfs_dw_job {
    const float *part;
    int64_t nslabs;
    int64_t per;          /* elements of one slab = K * Cs * Cg */
    int32_t K, A, B;      /* slab layout [K][A][B] */
    int32_t transpose;    /* dW[k][b][a] = sum part[.][k][a][b] instead of dW[k][a][b] */
    float *dW;
} wfs_dw_job;

int wfs_gather_dw(const int32_t *table, const int32_t *kmap_host, int32_t K, int32_t identity_k,
                  int64_t R, const void *S, int32_t Cs, const void *G, int64_t G_rows, int32_t Cg,
                  int32_t swap, float *dW, int32_t dtype, void *workspace, size_t workspace_bytes,
                  const int64_t *r_dev, wfs_dw_job *defer, int32_t packed_kl, void *stream);

/* torch.ops.spconv.indice_conv_backward as ONE call (round 4): dW[k] = X^T . dY[table[k]] (as wfs_gather_dw with
 * swap == 0) and dX = sum_k dY[table[k]] . W[k]^T (as wfs_gather_conv with transpose_w), both through the by-input
 * table of a conv / SubM layer (dense [K, R] or packed, packed_kl as above).  X [R, Cin] = the layer's input rows, dY
 * [dY_rows, Cout], W fp32 [K, Cin, Cout], dX [R, Cin], dW fp32 [K, Cin, Cout].  For 32 -> 32 layers with 16-bit rows
 * both products run in ONE launch (independent blocks of one grid: the second product no longer waits for the first
 * one's last block, and a kernel boundary of ~4.6 us inside a captured step goes); other shapes run the two entry
 * points one after the other.  Results are bit-identical to the separate calls.  workspace / defer as wfs_gather_dw. */
int wfs_conv_backward(const int32_t *table, int32_t K, int32_t identity_k, int64_t R, const void *X, const void *dY,
                      int64_t dY_rows, int32_t Cin, int32_t Cout, const float *W, void *dX, float *dW, int32_t dtype,
                      void *workspace, size_t workspace_bytes, const int64_t *r_dev, wfs_dw_job *defer,
                      int32_t packed_kl, void *stream);

/* Second stage of up to 16 deferred wfs_gather_dw calls in one launch (deterministic: fixed summation order). */
int wfs_dw_reduce_jobs(const wfs_dw_job *jobs, int32_t n, void *stream);

/* Conv bias gradient: out[c] = sum over the valid rows of X[r][c] (fp32 sums of fp32 / bf16 / fp16 rows; r_dev as
 * everywhere: NULL = R exact, else R is the capacity).  What autograd computes for spconv's `out_features += bias`
 * (reference layers with trainable_weights=True, src/models/SPConvBlocks.py:498).  Deterministic (fixed orders).      */
size_t wfs_column_sum_workspace_bytes(int32_t C);
int wfs_column_sum(const void *X, int64_t R, int32_t C, float *out, void *workspace, size_t workspace_bytes,
                   int32_t dtype, const int64_t *r_dev, void *stream);

/* Scatter form with fp32 atomics, used ONLY when the input holds duplicate coordinates (then
 * the inverse of a gather table is not a function):
 *     Y_accum[table[k, r], :] += X[r, :] . W[k]  (or W[k]^T)      Y_accum fp32, caller-initialised */
int wfs_scatter_conv(const int32_t *table, int32_t K, int32_t identity_k, int64_t R, const void *X,
                     int32_t Cx, const float *W, int32_t Cw_in, int32_t Cw_out, int32_t transpose_w,
                     float *Y_accum, int32_t dtype, void *stream);

/* BatchNorm1d (+ReLU) over the active rows ------------------------------------------------------
 * What spconv.SparseSequential does with the plain nn.BatchNorm1d / nn.ReLU modules the reference
 * puts after every sparse conv (src/models/SPConvBlocks.py:505-508): applied to .features [N, C],
 * statistics over the N active rows.  Two launches per direction (column reduction into per-block partials;
 * elementwise pass whose blocks first fold the partials in a fixed order): deterministic, no atomics.
 * training != 0: batch statistics (biased variance), running_mean/var (may be NULL) updated with
 * `momentum` using the unbiased variance and *num_batches_tracked (device int64, may be NULL)
 * incremented, exactly as torch.  training == 0: running statistics.
 * relu != 0 fuses y = max(0, .).  save_mean / save_invstd [C] are outputs the backward consumes.
 * gamma / beta may be NULL (affine=False).  C <= 1024.                                           */
size_t wfs_bn_workspace_bytes(int64_t N, int32_t C);


int wfs_bn_relu_fwd(const void *X, int64_t N, int32_t C, const float *gamma, const float *beta,
                    float *running_mean, float *running_var, int64_t *num_batches_tracked,
                    float momentum, float eps,
                    int32_t training, int32_t relu, void *Y, float *save_mean, float *save_invstd,
                    void *workspace, size_t workspace_bytes, int32_t dtype, const int64_t *n_dev,
                    void *stream);

int wfs_bn_relu_bwd(const void *X, const void *dY, int64_t N, int32_t C, const float *gamma,
                    const float *beta, const float *save_mean, const float *save_invstd,
                    int32_t training, int32_t relu, void *dX, float *dgamma, float *dbeta,
                    void *workspace, size_t workspace_bytes, int32_t dtype, const int64_t *n_dev,
                    void *stream);

/* After wfs_rulebook_emit of a regular conv whose site table was a direct grid: pointers into `workspace` to the
 * cell -> output row map of the build (ticket[cell] != 0xFFFFFFFF <=> the output cell b * out_volume + pos is active,
 * slot_id[cell] = its row).  Returns 1 and sets the pointers, or 0 when this build has no such map.  The map lives as
 * long as the workspace is kept. */
int wfs_rulebook_cell_map(const wfs_geometry *g, int64_t N, void *workspace, const uint32_t **ticket,
                          const int32_t **slot_id, int64_t *cells);

/* SparseConvTensor.dense() -------------------------------------------------------------------
 * Y is [B, C, *spatial] (channels first, contiguous) and must be zero-filled by the caller;
 * rows are assigned, not accumulated.  winner_ws: NULL when coordinates are unique, else int32
 * [B * volume] scratch that makes "the highest row wins" deterministic, as the reference's CPU
 * assignment order does (A.1).  wfs_to_dense_bwd gathers dX[m,c] = dY[b,c,pos] for every row. */
int wfs_to_dense(const void *X, const int32_t *indices, int64_t M, int32_t ndim,
                 const int32_t *spatial_host, int32_t batch_size, int32_t C, void *Y,
                 int32_t *winner_ws, int32_t dtype, const int64_t *m_dev, void *stream);

int wfs_to_dense_bwd(const void *dY, const int32_t *indices, int64_t M, int32_t ndim,
                     const int32_t *spatial_host, int32_t batch_size, int32_t C, void *dX,
                     int32_t dtype, const int64_t *m_dev, void *stream);

/* dense() and its backward through a cell -> row map (wfs_rulebook_cell_map of the conv that produced the rows, whose
 * coordinates are unique by construction).  Y [B, C, V] need NOT be zero-filled: every cell is written, a block owns 64
 * cells of one event and stores whole runs per channel.  V = out volume; C % 4 == 0, C <= 128, V even for 16-bit rows;
 * rows with id >= the valid count (m_dev) are treated as absent. */
int wfs_to_dense_mapped(const void *X, const uint32_t *ticket, const int32_t *slot_id, int64_t M,
                        const int64_t *m_dev, int32_t batch_size, int64_t V, int32_t C, void *Y, int32_t dtype,
                        void *stream);

int wfs_to_dense_bwd_mapped(const void *dY, const uint32_t *ticket, const int32_t *slot_id, int64_t M,
                            const int64_t *m_dev, int32_t batch_size, int64_t V, int32_t C, void *dX, int32_t dtype,
                            void *stream);

/* The head straight off the sparse rows (round 4; csrc/shead.hip).
 * Replaces the whole tail spconv.ToDense -> view(-1, n_linear) -> nn.Linear(n_linear, n_type) of the reference's
 * SPConvNet (src/models/SPConvNet.py:65-68; one-layer LinearBlock, src/models/ConvBlocks.py:82-102) AND its backward,
 * without materialising the dense tensor:
 *     Y[b][o]  = bias[o] + sum over rows i of event b, channels c:  X[i][c] * W[o][c * V + cell(i)]
 *     dX[i][c] = sum_o G[b(i)][o] * W[o][c * V + cell(i)];   dW[o][c * V + cell] = sum_b G[b][o] * X[row(b, cell)][c]
 * X [M, C] rows of the last conv's output (dtype), (ticket, slot_id) = that conv's cell -> row map (wfs_rulebook_cell_map,
 * or wfs_event_rulebook_conv's cell_row passed as both), V = its out volume, W fp32 [O, C * V] = nn.Linear.weight as it
 * is (channels first over the grid), Y / G fp32 [batch, O].  Cell-major kernels: a lane owns a cell, reads its weights
 * coalesced in the parameter's own layout and walks the events through the map.  O <= 4, C % 8 == 0, C <= 64
 * (wfs_sparse_head_ok).  dX / dW may be NULL; dB comes with dW.  defer as in wfs_gather_dw (the per-slice dW partials
 * then join the step's deferred slab reduction).  Deterministic (no atomics, fixed orders). */
int wfs_sparse_head_ok(int32_t batch, int64_t V, int32_t C, int32_t O, int32_t dtype);
size_t wfs_sparse_head_workspace_bytes(int32_t batch, int64_t V, int32_t C, int32_t O);
int wfs_sparse_head_fwd(const void *X, const uint32_t *ticket, const int32_t *slot_id, int64_t M,
                        const int64_t *m_dev, int32_t batch, int64_t V, int32_t C, const float *W,
                        const float *bias, int32_t O, float *Y, int32_t dtype, void *workspace,
                        size_t workspace_bytes, void *stream);
int wfs_sparse_head_bwd(const void *X, const float *G, const uint32_t *ticket, const int32_t *slot_id,
                        int64_t M, const int64_t *m_dev, int32_t batch, int64_t V, int32_t C, const float *W,
                        int32_t O, void *dX, float *dW, float *dB, int32_t dtype, void *workspace,
                        size_t workspace_bytes, wfs_dw_job *defer, void *stream);

/* classification head -----------------------------------------------------------------------
 * The reference flattens ToDense's output and applies the LinearBlock (src/models/SPConvNet.py:67-68,
 * src/models/ConvBlocks.py:82-102); the final nn.Linear has n_type = 2..4 outputs over tens of
 * thousands of inputs, a streaming problem rather than a GEMM.  X [B, I] fp32 or bf16 (dtype),
 * W [O, I] fp32 (nn.Linear.weight), Y / G [B, O] fp32, 1 <= O <= 8.  Rows with I % 8 == 0 and I >= 1024 stream
 * through vector kernels; any other row length (the short second layer of a two-layer head) takes scalar ones.
 *     Y = X W^T + bias;   dX = G W (X's dtype);   dW = G^T X (deterministic chunked reduction).
 * dX or dW may be NULL to skip that product; dB [O] (optional, computed with dW) = sum_b G[b][:].      */
size_t wfs_head_workspace_bytes(int64_t B, int64_t I, int32_t O);

int wfs_head_fwd(const void *X, int64_t B, int64_t I, const float *W, const float *bias, int32_t O,
                 float *Y, int32_t dtype, void *stream);

/* defer (optional): as in wfs_gather_dw -- when dX and dW are both asked for, the sum over the dW partials may be left
 * to a later wfs_dw_reduce_jobs (defer->nslabs > 0 on return; dW not written yet, the workspace must stay alive). */
int wfs_head_bwd(const void *X, const float *G, int64_t B, int64_t I, const float *W, int32_t O,
                 void *dX, float *dW, float *dB, int32_t dtype, void *workspace, size_t workspace_bytes,
                 wfs_dw_job *defer, void *stream);


/* hybrid front end -----------------------------------------------------------------------------------
 * TemporalConvNet(1, [1] * levels, kernel_size = k) as the reference's SPConvNet applies it to the waveform rows
 * before the sparse stack (src/models/SPConvNet.py:56-61,83-92; src/models/ConvBlocks.py:114-173): per level i two
 * causal k-tap FIR filters with dilation 2^i, ReLU after each, residual + ReLU.  One launch per direction; a row
 * [L] stays in LDS through all levels.  X, Y, dY, dX [N, L] fp32 or bf16 (dtype); taps [levels][2][k] and bias
 * [levels][2] are the EFFECTIVE filter taps (after weight norm) in DEVICE memory, fp32.  levels <= 8, k <= 8,
 * L <= 4096; the backward needs wfs_tcn_lds_bytes(L, levels, 1) <= 150 KiB (else WFS_EINVAL: use another path).
 * wfs_tcn_bwd writes per-row partial sums partial[N][levels][2][k + 1] (taps, then the bias); their sum over rows is
 * d loss / d (taps, bias).
 * Dropout (the nn.Dropout(p) after each of a level's two ReLUs, ConvBlocks.py:125-134): dropout_p in [0, 1) and
 * seed_dev = one int64 in DEVICE memory (drawn by the caller, e.g. torch.randint, so that a captured graph gets a
 * fresh seed per replay).  Kept elements are scaled by 1 / (1 - p); the decision for element (row, level, conv, t)
 * is a counter-based hash of the seed, so the backward pass reproduces the forward's masks from the same seed and
 * nothing is stored.  dropout_p == 0 (seed_dev may be NULL) is the eval-mode identity. */
size_t wfs_tcn_lds_bytes(int32_t L, int32_t levels, int32_t backward);

int wfs_tcn_fwd(const void *X, int64_t N, int32_t L, const float *taps, const float *bias, int32_t levels,
                int32_t k, void *Y, int32_t dtype, float dropout_p, const int64_t *seed_dev, void *stream);

int wfs_tcn_bwd(const void *X, const void *dY, int64_t N, int32_t L, const float *taps, const float *bias,
                int32_t levels, int32_t k, void *dX, float *partial, int32_t dtype, float dropout_p,
                const int64_t *seed_dev, void *stream);

/* Weight norm of the front end's taps (torch.nn.utils.weight_norm around every Conv1d of the reference's TemporalBlock,
 * src/models/ConvBlocks.py:118-131: w = g v / |v|), all convolutions in one launch each way.  param_ptrs: device array
 * of n_conv records of six device addresses {v [k], g [1], b [1] or 0, dv [k], dg [1], db [1] (each 0 = not wanted)},
 * convolution c = 2 * level + which.  wfs_tcn_taps_fwd fills taps [n_conv][k] and bias [n_conv] as wfs_tcn_fwd / _bwd
 * take them; wfs_tcn_taps_bwd sums wfs_tcn_bwd's partial [N][n_conv][k + 1] over the rows (fixed order) and writes
 * dv, dg, db.  n_conv <= 16, k <= 8.                                                                                  */
int wfs_tcn_taps_fwd(const void *param_ptrs, int32_t n_conv, int32_t k, float *taps, float *bias, void *stream);
int wfs_tcn_taps_bwd(const void *param_ptrs, int32_t n_conv, int32_t k, const float *partial, int64_t N, void *stream);

/* loss ---------------------------------------------------------------------------------------------
 * torch.nn.CrossEntropyLoss(reduction='mean') as the reference's LitPSD applies it to the [B, n_type] logits
 * (src/engineering/LitBase.py:38-43, LitPSD.py:102), forward AND d loss / d logits in one launch (torch runs six:
 * log_softmax, nll_loss, their backwards and two fills).  logits fp32 [B, C], target int64 [B]; rows whose target
 * equals ignore_index are skipped and do not count in the mean (torch's default is -100).
 * loss: device float [1]; dlogits [B, C] = (softmax - onehot) / counted rows, or NULL for the forward alone. */
int wfs_xent_mean_fwd_bwd(const float *logits, const int64_t *target, int64_t B, int32_t C,
                          int64_t ignore_index, float *loss, float *dlogits, void *stream);

/* optimizer ----------------------------------------------------------------------------------------
 * torch.optim.SGD's update (the optimizer of the reference's example configs, config/examples/GEP.json:51-69, built
 * by src/engineering/LitPSD.py:60-76) on one flat fp32 parameter buffer in a single launch; arithmetic and order of
 * torch/optim/sgd.py.  lr is read from device memory so that a scheduler can change it under a captured graph.
 * momentum_buf may be NULL when momentum == 0; first_step != 0 initialises it with the gradient, as torch does. */
int wfs_sgd_step(float *param, const float *grad, float *momentum_buf, int64_t n, const float *lr_dev,
                 float momentum, float dampening, float weight_decay, int32_t nesterov, int32_t first_step,
                 void *stream);

/* batch hand-over -------------------------------------------------------------------------------------
 * One launch that places a device-resident batch into the fixed buffers a captured step reads: coords [n, cols]
 * int32 (copied as is to coords_dst and, columns permuted by perm_host, to indices_dst -- the batch-first order the
 * reference produces at src/models/SPConvNet.py:64 -- either destination may be NULL), feats (feat_bytes bytes),
 * labels int64 [B], and the row count n into *n_valid_dst (may be NULL).  event_offsets (may be NULL): the
 * wfs_event_offsets table of the batch (`events` events, wfs_event_offsets_ints(events) ints; the batch index is the
 * source column that perm_host moves to the front), written by the same launch -- the captured step then starts with
 * the event-local rulebook build itself. */
int wfs_load_batch(const int32_t *coords, int64_t n, int32_t cols, const int32_t *perm_host, int32_t *coords_dst,
                   int32_t *indices_dst, const void *feats, void *feats_dst, int64_t feat_bytes,
                   const int64_t *labels, int64_t *labels_dst, int64_t B, int64_t *n_valid_dst,
                   int32_t *event_offsets, int32_t events, void *stream);

/* opt-in per-kernel timing (HIP events on the launch stream), used by bench.py's roofline ---- */
#define WFS_TIMER_GATHER_CONV 0
#define WFS_TIMER_GATHER_DW 1
#define WFS_TIMER_RULEBOOK 2
#define WFS_TIMER_CONV_BACKWARD 3   /* wfs_conv_backward's one-launch form (dW + dX) */
#define WFS_TIMER_COUNT 4
int wfs_timing_enable(int32_t on);                        /* also clears the table            */
int wfs_timing_read(int32_t timer, double *total_ms, int64_t *launches);  /* synchronises     */

#ifdef __cplusplus
}
#endif
#endif /* WFSPARSE_H */
