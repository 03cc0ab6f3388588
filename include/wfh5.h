/*
 * wfh5.h -- C ABI of libwfh5.so: native HDF5 -> sparse-COO reader for WaveformML's PSD datasets
 * (the step BEFORE the sparse-conv path).
 *
 * Replaces the h5py reads of reference src/datasets/HDF5Dataset.py:430-476 (_load_data: whole
 * columns into numpy) and the event-range slicing of :225-347 (_concat_range) for the two on-disk
 * layouts the PSD path uses:
 *
 *   WFH5_GROUP     "<table>/coord" int32 [n, 3|4], "<table>/waveform" [n, C] (int16 / float32),
 *                  optional "<table>/labels" [E], attribute "nevents" on the group; gzip-6 chunks
 *                  (written by reference src/datasets/PulseDataset.py:312-333 "combined" files)
 *   WFH5_COMPOUND  "<table>" = 1-D dataset of a compound type with members "coord" (int32[3|4]) and
 *                  "waveform" (int16[C] / float32[C]) among others (src/datasets/H5CompoundTypes.py:
 *                  105-120 WaveformPairCal; "WaveformPairs" / "Waveform3DPairs" tables,
 *                  src/datasets/PulseDataset.py:543-625), optional member "labels", attribute "nevents"
 *
 * Plain C types, caller-allocated HOST buffers (pin them for asynchronous H2D copies), no global state
 * besides a thread-local error string.  Thread-safe per handle (one handle per DataLoader worker).
 */
#ifndef WFH5_H
#define WFH5_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WFH5_OK 0
#define WFH5_EIO 1        /* file / object cannot be opened or read            */
#define WFH5_EFORMAT 2    /* the table is in neither supported layout          */
#define WFH5_EINVAL 3

#define WFH5_GROUP 0
#define WFH5_COMPOUND 1

typedef struct wfh5_file wfh5_file;

typedef struct wfh5_info {
    int64_t n_rows;          /* rows of coord / waveform                                   */
    int64_t n_events;        /* attribute "nevents" (-1 if absent)                         */
    int64_t n_labels;        /* length of the labels dataset / member (0 if absent)        */
    int32_t coord_cols;      /* 3 = (x, y, evt), 4 = (x, y, t, evt)                         */
    int32_t feat_cols;       /* C                                                          */
    int32_t feat_is_float;   /* waveform stored as float (else 16-bit ADC integers)        */
    int32_t layout;          /* WFH5_GROUP / WFH5_COMPOUND                                 */
} wfh5_info;

const char *wfh5_last_error(void);

int wfh5_open(const char *path, const char *table, wfh5_file **out);
/* The same with the names of the coordinate and feature members / datasets (the reference's datasets bind several:
 * "coord" / "waveform", "coord" / "pulse", "det" / "pulse", src/datasets/PulseDataset.py:700-1160).  Both names ""
 * opens a table for wfh5_read_member alone -- a label file (reference `label_file_pattern`,
 * src/datasets/HDF5Dataset.py:404-427), whose table need not hold coordinates at all. */
int wfh5_open_named(const char *path, const char *table, const char *coord_name, const char *feat_name,
                    wfh5_file **out);
void wfh5_close(wfh5_file *f);
int wfh5_get_info(const wfh5_file *f, wfh5_info *info);

/* rows [row0, row1): coords int32 [n, coord_cols]; feats float32 [n, feat_cols] = stored value * scale
 * (scale = 1/(2^14 - 1) reproduces reference HDF5Dataset.py:14-17,345-346 for ADC integers).
 * Either output pointer may be NULL.                                                              */
int wfh5_read_rows(wfh5_file *f, int64_t row0, int64_t row1, int32_t *coords, float *feats, float scale);

/* Worker threads for bulk reads of gzip-chunked tables (raw chunks are fetched serially, inflated and converted in
 * parallel; libhdf5's own filter pipeline is single-threaded).  Default: $WFH5_THREADS or 4.  Per process. */
int wfh5_set_threads(int n);

/* labels [e0, e1) widened to int64 (reference :319-327) */
int wfh5_read_labels(wfh5_file *f, int64_t e0, int64_t e1, int64_t *labels);

/* ANY member of a compound table / dataset of a group table by name -- the reference's per-row label columns
 * (`label_name: "PID"`, `"phys"`, `"EZ"`: config/examples/IoniClassifierCNN.json:75-89, SegQuantifier.json:70-78) and
 * `additional_fields` (src/datasets/HDF5Dataset.py:486-520): name NULL / "" = the FIRST member (what the reference
 * takes from a label file, :483).  wfh5_member_info: rows, array length per row, float or integer, stored element
 * size.  wfh5_read_member: rows [row0, row1) widened to float32 (as_float != 0) or int64, [n, cols] row-major. */
int wfh5_member_info(wfh5_file *f, const char *name, int64_t *rows, int32_t *cols, int32_t *is_float,
                     int32_t *elem_bytes);
int wfh5_read_member(wfh5_file *f, const char *name, int64_t row0, int64_t row1, int32_t as_float, void *out);

/* Row range of the events [e0, e1] (inclusive, as the reference's event_range): first row whose event id
 * (column event_col of coord) equals e0, and first row whose event id equals e1 + 1 (n_rows if e1 is the
 * last event) -- what `where(coords[:, c] == e)[0][0]` returns at reference :241-248.  Sorted files are searched by
 * bisection (a few chunk decodes) and the answer verified; unsorted ones fall back to a scan of the column.       */
int wfh5_event_rows(wfh5_file *f, int32_t event_col, int64_t e0, int64_t e1, int64_t *row0, int64_t *row1);

#ifdef __cplusplus
}
#endif
#endif /* WFH5_H */
