/* event_local.h -- C ABI of the event-local conv / conv-rulebook EXPERIMENTS (libwfs_evexp.so, built by the Makefile
 * here against libwfsparse.so; not part of the product).  See README.md. */
#ifndef WFS_EVENT_LOCAL_H
#define WFS_EVENT_LOCAL_H
#include "../../../include/wfsparse.h"
#ifdef __cplusplus
extern "C" {
#endif
/* wfs_slot_table: re-encodes a gather table [K, R] per event for wfs_event_conv: one 64-byte record per row of the
 *   table's row set, 32 uint16 slots, slot k = 1 + (table[mirror ? K-1-k : k][r] - first row of r's event in the
 *   gathered set), 0 = no neighbour; identity_k >= 0: the row itself.  slots: R * 64 bytes.
 * wfs_event_conv: Y[r] = bias + sum_k X[table[mirror ? K-1-k : k, r]] . W[k] (^T if transpose_w) for 32 -> 32
 *   channels, 16-bit rows, K <= 27 (wfs_event_conv_ok), ONE WORKGROUP PER EVENT, the event's input rows staged in LDS.
 *   out_events / in_events: offsets (wfs_event_offsets) of the row set of Y and of X.  slots_mirror = 1: records in the
 *   table's own offset order, read mirrored (K = 27: the SubM forward through wfs_event_rulebook_subm's records).
 *   Events whose input rows exceed the LDS capacity (~1350 rows) and row sets not grouped by event gather from X
 *   through `table` inside the same launch.  Same arithmetic and summation order as wfs_gather_conv (bit-equal). */
int wfs_slot_table(const int32_t *table, int32_t mirror, int32_t K, int32_t identity_k, int64_t R,
                   const int32_t *out_events, const int32_t *in_events, int32_t batch_size, const int64_t *r_dev,
                   void *slots, void *stream);
int wfs_event_conv_ok(int32_t K, int32_t Cx, int32_t Cw_in, int32_t Cw_out, int32_t dtype, int32_t batch_size);
int wfs_event_conv(const int32_t *table, int32_t mirror, int32_t K, int32_t identity_k, int64_t R, const void *slots,
                   int32_t slots_mirror, const int32_t *out_events, const int32_t *in_events, int32_t batch_size,
                   const void *X, const float *W, int32_t transpose_w, const float *bias, void *Y, int32_t dtype,
                   const int64_t *r_dev, void *stream);
/* Regular / strided conv rulebook (ndim <= 3, kernel 3 in every dim): the whole of wfs_rulebook_plan + _emit in two
 * launches (COUNT, EMIT), one workgroup per event; bit-identical tables; flags int32 [4] zeroed by the caller. */
int wfs_event_rulebook_conv_ok(const wfs_geometry *g);
size_t wfs_event_rulebook_conv_workspace_bytes(int32_t batch_size);
int wfs_event_rulebook_conv(const wfs_geometry *g, const int32_t *indices, int64_t N, const int64_t *n_dev,
                            const int32_t *in_events, int32_t *nbr_out, int32_t *nbr_in, int32_t *out_indices,
                            int64_t M_cap, int32_t *out_events, int64_t *info, int64_t *m_dev, int32_t *overflow_dev,
                            int32_t *flags, uint32_t *cell_ticket, int32_t *cell_row, void *slots_bwd, void *slots_fwd,
                            void *workspace, size_t workspace_bytes, void *stream);
#ifdef __cplusplus
}
#endif
#endif
