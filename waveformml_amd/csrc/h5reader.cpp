// h5reader.cpp -- libwfh5.so: native HDF5 -> COO reader (include/wfh5.h).  Host-only C++ over the HDF5 C API
// (libhdf5 1.10 from the image's /opt/conda; gzip chunk decoding is libhdf5's).
//
// Replaces the h5py column reads + event slicing of reference src/datasets/HDF5Dataset.py:225-347,430-476 for the
// group layout written by src/datasets/PulseDataset.py:312-333 and the compound tables of
// src/datasets/H5CompoundTypes.py:105-120.  Only the members the PSD path consumes (coord, waveform, labels) are
// materialised: compound records are read through a memory type that names just those members.
#include <hdf5.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <unistd.h>
#include <libdeflate.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/wfh5.h"

static thread_local char g_err[512] = "";
static void set_err(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *wfh5_last_error(void) { return g_err; }

struct wfh5_file {
    hid_t file = -1;
    int fd = -1;                  // the same file through POSIX, for parallel reads of raw chunks at known addresses
    int layout = WFH5_GROUP;
    // group layout
    hid_t d_coord = -1, d_feat = -1, d_labels = -1;
    // compound layout
    hid_t d_table = -1;
    bool labels_member = false;
    std::string coord_name = "coord", feat_name = "waveform";      // member / dataset names bound at open
    hid_t group = -1;               // group layout: the table's group, for wfh5_read_member
    wfh5_info info{};
    std::vector<int32_t> event_col_cache;     // the event-id column, loaded once for the linear event search
    int cached_event_col = -1;
    // chunk-parallel fast path (gzip-only chunked storage whose chunks span whole rows): raw chunks are fetched with
    // H5Dread_chunk (serial, cheap) and inflated + converted on worker threads -- libhdf5's own filter pipeline is
    // single-threaded and is what bounds a plain H5Dread of these files.
    struct Fast {
        bool ok = false;
        hid_t dset = -1;
        hsize_t chunk_rows = 0;
        size_t row_bytes = 0;          // bytes of one row / record as stored
        size_t off = 0;                // byte offset of the member inside the record (compound) or 0
        size_t elem = 0;               // stored element size
        bool is_float = false;
        int cols = 0;
        bool two_d = false;
    };
    Fast fast_coord, fast_feat;
};

static int g_threads = -1;
static int reader_threads() {
    if (g_threads < 0) {
        const char *e = getenv("WFH5_THREADS");
        g_threads = e ? atoi(e) : 4;
        if (g_threads < 1) g_threads = 1;
    }
    return g_threads;
}
extern "C" int wfh5_set_threads(int n) {
    g_threads = n < 1 ? 1 : n;
    return WFH5_OK;
}

static bool little_endian_native(hid_t t) { return H5Tget_size(t) == 1 || H5Tget_order(t) == H5T_ORDER_LE; }

// is `dset` stored in gzip-only chunks of whole rows?  (1-D compound table, or 2-D [n, cols] with chunk (r, cols))
static bool chunked_gzip_rows(hid_t dset, hsize_t *chunk_rows) {
    hid_t pl = H5Dget_create_plist(dset);
    bool ok = false;
    if (pl >= 0 && H5Pget_layout(pl) == H5D_CHUNKED && H5Pget_nfilters(pl) == 1) {
        unsigned flags = 0, cfg = 0;
        size_t nelm = 0;
        char name[16];
        if (H5Pget_filter2(pl, 0, &flags, &nelm, nullptr, sizeof(name), name, &cfg) == H5Z_FILTER_DEFLATE) {
            hid_t sp = H5Dget_space(dset);
            int nd = H5Sget_simple_extent_ndims(sp);
            hsize_t dims[2] = {0, 1}, cd[2] = {0, 1};
            if (nd >= 1 && nd <= 2) {
                H5Sget_simple_extent_dims(sp, dims, nullptr);
                H5Pget_chunk(pl, nd, cd);
                ok = cd[0] >= 1 && (nd == 1 || cd[1] == dims[1]);
                *chunk_rows = cd[0];
            }
            H5Sclose(sp);
        }
    }
    if (pl >= 0) H5Pclose(pl);
    return ok;
}


static int64_t read_nevents(hid_t obj) {
    if (H5Aexists(obj, "nevents") <= 0) return -1;
    hid_t a = H5Aopen(obj, "nevents", H5P_DEFAULT);
    if (a < 0) return -1;
    int64_t v = -1;
    hid_t sp = H5Aget_space(a);
    hssize_t n = H5Sget_simple_extent_npoints(sp);
    std::vector<int64_t> buf(n > 0 ? n : 1, -1);
    if (H5Aread(a, H5T_NATIVE_INT64, buf.data()) >= 0) v = buf[0];
    H5Sclose(sp);
    H5Aclose(a);
    return v;
}

static bool dims2(hid_t dset, hsize_t *d0, hsize_t *d1) {
    hid_t sp = H5Dget_space(dset);
    int nd = H5Sget_simple_extent_ndims(sp);
    hsize_t dims[4] = {0, 1, 1, 1};
    if (nd >= 1 && nd <= 4) H5Sget_simple_extent_dims(sp, dims, nullptr);
    H5Sclose(sp);
    if (nd < 1 || nd > 2) return false;
    *d0 = dims[0];
    *d1 = nd == 2 ? dims[1] : 1;
    return true;
}

extern "C" void wfh5_close(wfh5_file *f) {
    if (!f) return;
    if (f->group >= 0) H5Gclose(f->group);
    if (f->d_coord >= 0) H5Dclose(f->d_coord);
    if (f->d_feat >= 0) H5Dclose(f->d_feat);
    if (f->d_labels >= 0) H5Dclose(f->d_labels);
    if (f->d_table >= 0) H5Dclose(f->d_table);
    if (f->file >= 0) H5Fclose(f->file);
    if (f->fd >= 0) close(f->fd);
    delete f;
}

// number of elements of an array-typed (or scalar) compound member
static hsize_t member_len(hid_t mtype) {
    if (H5Tget_class(mtype) != H5T_ARRAY) return 1;
    int nd = H5Tget_array_ndims(mtype);
    hsize_t dims[8];
    H5Tget_array_dims2(mtype, dims);
    hsize_t n = 1;
    for (int i = 0; i < nd; ++i) n *= dims[i];
    return n;
}

extern "C" int wfh5_open_named(const char *path, const char *table, const char *coord_name, const char *feat_name,
                               wfh5_file **out) {
    if (!path || !table || !out) {
        set_err("NULL argument");
        return WFH5_EINVAL;
    }
    H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);          // errors are reported through return codes
    wfh5_file *f = new wfh5_file;
    if (coord_name) f->coord_name = coord_name;
    if (feat_name) f->feat_name = feat_name;
    const bool members_only = f->coord_name.empty() && f->feat_name.empty();      // e.g. a label file: wfh5_read_member only
    f->file = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f->file < 0) {
        set_err("cannot open %s", path);
        wfh5_close(f);
        return WFH5_EIO;
    }
    f->fd = open(path, O_RDONLY);
    H5O_info_t oi;
    if (H5Lexists(f->file, table, H5P_DEFAULT) <= 0 || H5Oget_info_by_name(f->file, table, &oi, H5P_DEFAULT) < 0) {
        set_err("%s: no object named %s", path, table);
        wfh5_close(f);
        return WFH5_EFORMAT;
    }
    if (oi.type == H5O_TYPE_GROUP) {
        f->layout = WFH5_GROUP;
        hid_t g = H5Gopen2(f->file, table, H5P_DEFAULT);
        f->info.n_events = read_nevents(g);
        const char *cn = f->coord_name.c_str(), *fn = f->feat_name.c_str();
        f->d_coord = (!members_only && H5Lexists(g, cn, H5P_DEFAULT) > 0) ? H5Dopen2(g, cn, H5P_DEFAULT) : -1;
        f->d_feat = (!members_only && H5Lexists(g, fn, H5P_DEFAULT) > 0) ? H5Dopen2(g, fn, H5P_DEFAULT) : -1;
        f->d_labels = H5Lexists(g, "labels", H5P_DEFAULT) > 0 ? H5Dopen2(g, "labels", H5P_DEFAULT) : -1;
        f->group = g;
        if (members_only) {
            f->info.layout = f->layout;
            *out = f;
            return WFH5_OK;
        }
        hsize_t n0, c0, n1, c1;
        if (f->d_coord < 0 || f->d_feat < 0 || !dims2(f->d_coord, &n0, &c0) || !dims2(f->d_feat, &n1, &c1) || n0 != n1) {
            set_err("%s:%s is a group without matching %s / %s datasets", path, table, cn, fn);
            wfh5_close(f);
            return WFH5_EFORMAT;
        }
        f->info.n_rows = (int64_t)n0;
        f->info.coord_cols = (int32_t)c0;
        f->info.feat_cols = (int32_t)c1;
        hid_t t = H5Dget_type(f->d_feat);
        f->info.feat_is_float = H5Tget_class(t) == H5T_FLOAT;
        {
            hid_t tcd = H5Dget_type(f->d_coord);
            hsize_t cr = 0;
            if (H5Tget_class(tcd) == H5T_INTEGER && H5Tget_size(tcd) == 4 && little_endian_native(tcd) &&
                chunked_gzip_rows(f->d_coord, &cr)) {
                f->fast_coord.ok = true;
                f->fast_coord.dset = f->d_coord;
                f->fast_coord.chunk_rows = cr;
                f->fast_coord.row_bytes = (size_t)c0 * 4;
                f->fast_coord.elem = 4;
                f->fast_coord.cols = (int)c0;
                f->fast_coord.two_d = true;
            }
            H5Tclose(tcd);
            const size_t es = H5Tget_size(t);
            const bool flt = H5Tget_class(t) == H5T_FLOAT;
            if (((flt && es == 4) || (!flt && es == 2 && H5Tget_sign(t) == H5T_SGN_2)) && little_endian_native(t) &&
                chunked_gzip_rows(f->d_feat, &cr)) {
                f->fast_feat.ok = true;
                f->fast_feat.dset = f->d_feat;
                f->fast_feat.chunk_rows = cr;
                f->fast_feat.row_bytes = (size_t)c1 * es;
                f->fast_feat.elem = es;
                f->fast_feat.is_float = flt;
                f->fast_feat.cols = (int)c1;
                f->fast_feat.two_d = true;
            }
        }
        H5Tclose(t);
        if (f->d_labels >= 0) {
            hsize_t nl, cl;
            f->info.n_labels = dims2(f->d_labels, &nl, &cl) ? (int64_t)nl : 0;
        }
    } else if (oi.type == H5O_TYPE_DATASET) {
        f->layout = WFH5_COMPOUND;
        f->d_table = H5Dopen2(f->file, table, H5P_DEFAULT);
        hid_t t = f->d_table >= 0 ? H5Dget_type(f->d_table) : -1;
        hsize_t n0 = 0, c0 = 0;
        if (t < 0 || H5Tget_class(t) != H5T_COMPOUND || !dims2(f->d_table, &n0, &c0)) {
            set_err("%s:%s is not a compound table", path, table);
            if (t >= 0) H5Tclose(t);
            wfh5_close(f);
            return WFH5_EFORMAT;
        }
        if (members_only) {
            H5Tclose(t);
            f->info.n_rows = (int64_t)n0;
            f->info.n_events = read_nevents(f->d_table);
            f->info.layout = f->layout;
            *out = f;
            return WFH5_OK;
        }
        int ic = H5Tget_member_index(t, f->coord_name.c_str()), iw = H5Tget_member_index(t, f->feat_name.c_str());
        if (ic < 0 || iw < 0) {
            set_err("%s:%s has no %s / %s members", path, table, f->coord_name.c_str(), f->feat_name.c_str());
            H5Tclose(t);
            wfh5_close(f);
            return WFH5_EFORMAT;
        }
        hid_t tc = H5Tget_member_type(t, ic), tw = H5Tget_member_type(t, iw);
        f->info.coord_cols = (int32_t)member_len(tc);
        f->info.feat_cols = (int32_t)member_len(tw);
        hid_t twb = H5Tget_class(tw) == H5T_ARRAY ? H5Tget_super(tw) : H5Tcopy(tw);
        f->info.feat_is_float = H5Tget_class(twb) == H5T_FLOAT;
        {
            hid_t tcb = H5Tget_class(tc) == H5T_ARRAY ? H5Tget_super(tc) : H5Tcopy(tc);
            hsize_t cr = 0;
            const bool chunked = chunked_gzip_rows(f->d_table, &cr);
            const size_t rec = H5Tget_size(t);
            if (chunked && H5Tget_class(tcb) == H5T_INTEGER && H5Tget_size(tcb) == 4 && little_endian_native(tcb)) {
                f->fast_coord.ok = true;
                f->fast_coord.dset = f->d_table;
                f->fast_coord.chunk_rows = cr;
                f->fast_coord.row_bytes = rec;
                f->fast_coord.off = H5Tget_member_offset(t, (unsigned)ic);
                f->fast_coord.elem = 4;
                f->fast_coord.cols = f->info.coord_cols;
            }
            const size_t es = H5Tget_size(twb);
            const bool flt = H5Tget_class(twb) == H5T_FLOAT;
            if (chunked && ((flt && es == 4) || (!flt && es == 2 && H5Tget_sign(twb) == H5T_SGN_2)) &&
                little_endian_native(twb)) {
                f->fast_feat.ok = true;
                f->fast_feat.dset = f->d_table;
                f->fast_feat.chunk_rows = cr;
                f->fast_feat.row_bytes = rec;
                f->fast_feat.off = H5Tget_member_offset(t, (unsigned)iw);
                f->fast_feat.elem = es;
                f->fast_feat.is_float = flt;
                f->fast_feat.cols = f->info.feat_cols;
            }
            H5Tclose(tcb);
        }
        H5Tclose(twb);
        H5Tclose(tc);
        H5Tclose(tw);
        f->labels_member = H5Tget_member_index(t, "labels") >= 0;
        H5Tclose(t);
        f->info.n_rows = (int64_t)n0;
        f->info.n_events = read_nevents(f->d_table);
        f->info.n_labels = f->labels_member ? (int64_t)n0 : 0;
    } else {
        set_err("%s:%s is neither a group nor a dataset", path, table);
        wfh5_close(f);
        return WFH5_EFORMAT;
    }
    f->info.layout = f->layout;
    *out = f;
    return WFH5_OK;
}

extern "C" int wfh5_get_info(const wfh5_file *f, wfh5_info *info) {
    if (!f || !info) {
        set_err("NULL argument");
        return WFH5_EINVAL;
    }
    *info = f->info;
    return WFH5_OK;
}

// hyperslab read of rows [r0, r1) x all columns of a 1-D / 2-D dataset into `buf` of memory type `mt`
static int read_slab(hid_t dset, hsize_t r0, hsize_t r1, hsize_t cols, bool two_d, hid_t mt, void *buf) {
    hid_t fs = H5Dget_space(dset);
    hsize_t start[2] = {r0, 0}, count[2] = {r1 - r0, cols};
    H5Sselect_hyperslab(fs, H5S_SELECT_SET, start, nullptr, count, nullptr);
    hid_t ms = H5Screate_simple(two_d ? 2 : 1, count, nullptr);
    herr_t rc = H5Dread(dset, mt, ms, fs, H5P_DEFAULT, buf);
    H5Sclose(ms);
    H5Sclose(fs);
    return rc < 0 ? WFH5_EIO : WFH5_OK;
}

// read ONE array member of the compound table for rows [r0, r1) as `base` elements (HDF5 converts)
static int read_member(wfh5_file *f, const char *name, hsize_t len, hid_t base, size_t base_size, hsize_t r0, hsize_t r1,
                       void *buf) {
    hid_t arr = len > 1 ? H5Tarray_create2(base, 1, &len) : H5Tcopy(base);
    hid_t mt = H5Tcreate(H5T_COMPOUND, base_size * len);
    H5Tinsert(mt, name, 0, arr);
    int rc = read_slab(f->d_table, r0, r1, 1, false, mt, buf);
    H5Tclose(mt);
    H5Tclose(arr);
    return rc;
}

// rows [r0, r1) of one or two members that live in the same chunked dataset (compound) or of one 2-D dataset:
// chunk addresses from libhdf5 (serial), then pread + inflate + convert on worker threads.  Returns false if anything is unexpected (the caller then
// uses H5Dread).
static bool read_rows_chunked(int fd, const wfh5_file::Fast *fc, int32_t *coords, const wfh5_file::Fast *ff,
                              float *feats, float scale, hsize_t r0, hsize_t r1) {
    const wfh5_file::Fast *any = fc ? fc : ff;
    const hsize_t cr = any->chunk_rows;
    const hsize_t c_first = r0 / cr, c_last = (r1 - 1) / cr;
    const size_t nchunks = (size_t)(c_last - c_first + 1);
    // where the raw chunks are (libhdf5's chunk index, serial); their bytes are then read with pread() on the worker
    // threads -- H5Dread_chunk costs ~0.5 ms per call and would serialise the whole read
    std::vector<haddr_t> addr(nchunks);
    std::vector<hsize_t> size(nchunks);
    for (size_t i = 0; i < nchunks; ++i) {
        hsize_t off[2] = {(c_first + i) * cr, 0};
        unsigned mask = 0;
        if (H5Dget_chunk_info_by_coord(any->dset, off, &mask, &addr[i], &size[i]) < 0 || size[i] == 0 || mask != 0 ||
            addr[i] == HADDR_UNDEF)
            return false;
    }
    if (fd < 0) return false;
    std::atomic<size_t> next(0);
    std::atomic<bool> failed(false);
    auto work = [&]() {
        std::vector<unsigned char> rows((size_t)cr * any->row_bytes), raw;
        struct Inflater {              // freed when the worker returns
            libdeflate_decompressor *d = nullptr;
            ~Inflater() { if (d) libdeflate_free_decompressor(d); }
        } guard;
        libdeflate_decompressor *&inflater = guard.d;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= nchunks || failed.load()) return;
            raw.resize((size_t)size[i]);
            size_t done = 0;
            while (done < raw.size()) {
                ssize_t n = pread(fd, raw.data() + done, raw.size() - done, (off_t)(addr[i] + done));
                if (n <= 0) {
                    failed.store(true);
                    return;
                }
                done += (size_t)n;
            }
            // libdeflate's one-shot inflate of the zlib stream (2 - 3 x zlib's uncompress on these records; a worker's
            // time is 70 % inflate: tools/soak_from_files.py); one decompressor per worker thread
            size_t got = 0;
            if (!inflater) inflater = libdeflate_alloc_decompressor();
            if (!inflater || libdeflate_zlib_decompress(inflater, raw.data(), raw.size(), rows.data(), rows.size(), &got) !=
                                 LIBDEFLATE_SUCCESS || got != rows.size()) {
                failed.store(true);
                return;
            }
            const hsize_t base = (c_first + i) * cr;
            const hsize_t a = std::max(base, r0), b = std::min(base + cr, r1);
            for (hsize_t r = a; r < b; ++r) {
                const unsigned char *rec = rows.data() + (size_t)(r - base) * any->row_bytes;
                if (fc && coords) memcpy(coords + (size_t)(r - r0) * fc->cols, rec + fc->off, (size_t)fc->cols * 4);
                if (ff && feats) {
                    float *o = feats + (size_t)(r - r0) * ff->cols;
                    if (ff->is_float) {
                        const float *src = (const float *)(rec + ff->off);
                        for (int c = 0; c < ff->cols; ++c) o[c] = src[c] * scale;
                    } else {
                        const int16_t *src = (const int16_t *)(rec + ff->off);
                        for (int c = 0; c < ff->cols; ++c) o[c] = (float)src[c] * scale;
                    }
                }
            }
        }
    };
    const int nt = (int)std::min<size_t>((size_t)reader_threads(), nchunks);
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    return !failed.load();
}

extern "C" int wfh5_read_rows(wfh5_file *f, int64_t row0, int64_t row1, int32_t *coords, float *feats, float scale) {
    if (!f || row0 < 0 || row1 < row0 || row1 > f->info.n_rows) {
        set_err("bad row range [%lld, %lld) of %lld", (long long)row0, (long long)row1, f ? (long long)f->info.n_rows : -1ll);
        return WFH5_EINVAL;
    }
    if (row1 == row0) return WFH5_OK;
    int rc = WFH5_OK;
    const hsize_t r0 = (hsize_t)row0, r1 = (hsize_t)row1;
    // chunk-parallel path first (bulk reads only: single rows -- the event search's probes -- go through H5Dread, whose
    // chunk cache keeps the probed chunk); it scales the features itself, so the generic scaling below is skipped
    const bool bulk = row1 - row0 >= 64;
    if (!bulk) {
    } else if (f->layout == WFH5_COMPOUND && (!coords || f->fast_coord.ok) && (!feats || f->fast_feat.ok) && (coords || feats)) {
        if (read_rows_chunked(f->fd, coords ? &f->fast_coord : nullptr, coords, feats ? &f->fast_feat : nullptr, feats, scale, r0,
                              r1))
            return WFH5_OK;
    } else if (f->layout == WFH5_GROUP) {
        bool done_c = !coords, done_f = !feats;
        if (coords && f->fast_coord.ok) done_c = read_rows_chunked(f->fd, &f->fast_coord, coords, nullptr, nullptr, 1.0f, r0, r1);
        if (feats && f->fast_feat.ok) done_f = read_rows_chunked(f->fd, nullptr, nullptr, &f->fast_feat, feats, scale, r0, r1);
        if (done_c && done_f) return WFH5_OK;
        if (done_c) coords = nullptr;                 // the rest goes through H5Dread
        if (done_f) feats = nullptr;
    }
    if (f->layout == WFH5_GROUP) {
        if (coords) rc = read_slab(f->d_coord, r0, r1, f->info.coord_cols, true, H5T_NATIVE_INT32, coords);
        if (rc == WFH5_OK && feats) rc = read_slab(f->d_feat, r0, r1, f->info.feat_cols, true, H5T_NATIVE_FLOAT, feats);
    } else {
        if (coords) rc = read_member(f, f->coord_name.c_str(), f->info.coord_cols, H5T_NATIVE_INT32, 4, r0, r1, coords);
        if (rc == WFH5_OK && feats)
            rc = read_member(f, f->feat_name.c_str(), f->info.feat_cols, H5T_NATIVE_FLOAT, 4, r0, r1, feats);
    }
    if (rc != WFH5_OK) {
        set_err("HDF5 read failed for rows [%lld, %lld)", (long long)row0, (long long)row1);
        return rc;
    }
    if (feats && scale != 1.0f) {
        const size_t n = (size_t)(row1 - row0) * f->info.feat_cols;
        for (size_t i = 0; i < n; ++i) feats[i] *= scale;
    }
    return WFH5_OK;
}

extern "C" int wfh5_read_labels(wfh5_file *f, int64_t e0, int64_t e1, int64_t *labels) {
    if (!f || !labels || e0 < 0 || e1 < e0 || e1 > f->info.n_labels) {
        set_err("bad label range");
        return WFH5_EINVAL;
    }
    if (e1 == e0) return WFH5_OK;
    int rc;
    if (f->layout == WFH5_GROUP) {
        if (f->d_labels < 0) {
            set_err("no labels dataset");
            return WFH5_EFORMAT;
        }
        rc = read_slab(f->d_labels, (hsize_t)e0, (hsize_t)e1, 1, false, H5T_NATIVE_INT64, labels);
    } else {
        if (!f->labels_member) {
            set_err("no labels member");
            return WFH5_EFORMAT;
        }
        rc = read_member(f, "labels", 1, H5T_NATIVE_INT64, 8, (hsize_t)e0, (hsize_t)e1, labels);
    }
    if (rc != WFH5_OK) set_err("HDF5 label read failed");
    return rc;
}

extern "C" int wfh5_event_rows(wfh5_file *f, int32_t event_col, int64_t e0, int64_t e1, int64_t *row0, int64_t *row1) {
    if (!f || !row0 || !row1 || event_col < 0 || event_col >= f->info.coord_cols || e1 < e0) {
        set_err("bad argument");
        return WFH5_EINVAL;
    }
    const int64_t n = f->info.n_rows;
    // Sorted files (what the simulation chain writes, and what collate_fn's event re-numbering needs): two binary
    // searches with single-row reads -- a handful of chunk decodes instead of the whole column.  The result is checked
    // against the first-occurrence rule at both ends; anything inconsistent falls back to the linear scan below.
    if (f->cached_event_col != event_col && n > 0) {
        std::vector<int32_t> one((size_t)f->info.coord_cols);
        bool io_ok = true;
        auto ev = [&](int64_t r) -> int64_t {
            if (wfh5_read_rows(f, r, r + 1, one.data(), nullptr, 1.0f) != WFH5_OK) {
                io_ok = false;
                return 0;
            }
            return one[(size_t)event_col];
        };
        auto lower = [&](int64_t e) {
            int64_t lo = 0, hi = n;
            while (lo < hi && io_ok) {
                int64_t mid = lo + (hi - lo) / 2;
                if (ev(mid) < e) lo = mid + 1; else hi = mid;
            }
            return lo;
        };
        const int64_t a = e0 > 0 ? lower(e0) : 0, b = lower(e1 + 1);
        bool good = io_ok && a < n && ev(a) == e0 && (a == 0 || ev(a - 1) < e0) && a < b && ev(b - 1) <= e1 &&
                    (b == n || ev(b) == e1 + 1);
        if (e0 == 0) good = good && ev(0) >= 0;
        if (good && io_ok) {
            *row0 = a;
            *row1 = b;
            return WFH5_OK;
        }
    }
    if (f->cached_event_col != event_col) {
        std::vector<int32_t> all((size_t)n * f->info.coord_cols);
        int rc = n ? wfh5_read_rows(f, 0, n, all.data(), nullptr, 1.0f) : WFH5_OK;
        if (rc != WFH5_OK) return rc;
        f->event_col_cache.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) f->event_col_cache[(size_t)i] = all[(size_t)i * f->info.coord_cols + event_col];
        f->cached_event_col = event_col;
    }
    // first occurrence, scanning like the reference's `where(col == e)[0][0]` (no sortedness assumed)
    auto first_of = [&](int64_t e, int64_t from) -> int64_t {
        for (int64_t i = from; i < n; ++i)
            if (f->event_col_cache[(size_t)i] == e) return i;
        return -1;
    };
    int64_t a = e0 > 0 ? first_of(e0, 0) : 0;
    if (a < 0) {
        set_err("event %lld not found", (long long)e0);
        return WFH5_EINVAL;
    }
    int64_t b = first_of(e1 + 1, 0);
    *row0 = a;
    *row1 = b < 0 ? n : b;
    return WFH5_OK;
}

extern "C" int wfh5_open(const char *path, const char *table, wfh5_file **out) {
    return wfh5_open_named(path, table, "coord", "waveform", out);
}

// the member (compound layout) or dataset (group layout) called `name`; NULL / "" = the first member of a compound
static int find_member(wfh5_file *f, const char *name, std::string *resolved, hid_t *dset, hsize_t *rows, hsize_t *cols,
                       bool *is_float, size_t *elem) {
    *dset = -1;
    if (f->layout == WFH5_COMPOUND) {
        hid_t t = H5Dget_type(f->d_table);
        int idx = (name && name[0]) ? H5Tget_member_index(t, name) : 0;
        if (idx < 0 || idx >= H5Tget_nmembers(t)) {
            H5Tclose(t);
            set_err("no member named %s", name ? name : "(first)");
            return WFH5_EFORMAT;
        }
        char *mn = H5Tget_member_name(t, (unsigned)idx);
        *resolved = mn;
        H5free_memory(mn);
        hid_t mt = H5Tget_member_type(t, (unsigned)idx);
        *cols = member_len(mt);
        hid_t base = H5Tget_class(mt) == H5T_ARRAY ? H5Tget_super(mt) : H5Tcopy(mt);
        const H5T_class_t cls = H5Tget_class(base);
        *is_float = cls == H5T_FLOAT;
        *elem = H5Tget_size(base);
        H5Tclose(base);
        H5Tclose(mt);
        H5Tclose(t);
        if (cls != H5T_FLOAT && cls != H5T_INTEGER) {
            set_err("member %s is neither integer nor float", resolved->c_str());
            return WFH5_EFORMAT;
        }
        hsize_t n0 = 0, c0 = 0;
        dims2(f->d_table, &n0, &c0);
        *rows = n0;
        return WFH5_OK;
    }
    if (!name || !name[0] || f->group < 0 || H5Lexists(f->group, name, H5P_DEFAULT) <= 0) {
        set_err("no dataset named %s in the group", name ? name : "(none)");
        return WFH5_EFORMAT;
    }
    *resolved = name;
    *dset = H5Dopen2(f->group, name, H5P_DEFAULT);
    hsize_t n0 = 0, c0 = 0;
    if (*dset < 0 || !dims2(*dset, &n0, &c0)) {
        set_err("dataset %s cannot be read as [n] / [n, c]", name);
        if (*dset >= 0) H5Dclose(*dset);
        return WFH5_EFORMAT;
    }
    hid_t t = H5Dget_type(*dset);
    *is_float = H5Tget_class(t) == H5T_FLOAT;
    *elem = H5Tget_size(t);
    H5Tclose(t);
    *rows = n0;
    *cols = c0;
    return WFH5_OK;
}

extern "C" int wfh5_member_info(wfh5_file *f, const char *name, int64_t *rows, int32_t *cols, int32_t *is_float,
                                int32_t *elem_bytes) {
    if (!f) {
        set_err("NULL handle");
        return WFH5_EINVAL;
    }
    std::string rn;
    hid_t d;
    hsize_t n, c;
    bool fl;
    size_t es;
    int rc = find_member(f, name, &rn, &d, &n, &c, &fl, &es);
    if (rc != WFH5_OK) return rc;
    if (d >= 0) H5Dclose(d);
    if (rows) *rows = (int64_t)n;
    if (cols) *cols = (int32_t)c;
    if (is_float) *is_float = fl ? 1 : 0;
    if (elem_bytes) *elem_bytes = (int32_t)es;
    return WFH5_OK;
}

extern "C" int wfh5_read_member(wfh5_file *f, const char *name, int64_t row0, int64_t row1, int32_t as_float, void *out) {
    if (!f || !out || row0 < 0 || row1 < row0) {
        set_err("bad argument");
        return WFH5_EINVAL;
    }
    std::string rn;
    hid_t d;
    hsize_t n, c;
    bool fl;
    size_t es;
    int rc = find_member(f, name, &rn, &d, &n, &c, &fl, &es);
    if (rc != WFH5_OK) return rc;
    if ((hsize_t)row1 > n) {
        if (d >= 0) H5Dclose(d);
        set_err("rows [%lld, %lld) of %s beyond %lld", (long long)row0, (long long)row1, rn.c_str(), (long long)n);
        return WFH5_EINVAL;
    }
    if (row1 == row0) {
        if (d >= 0) H5Dclose(d);
        return WFH5_OK;
    }
    const hid_t base = as_float ? H5T_NATIVE_FLOAT : H5T_NATIVE_INT64;
    const size_t bs = as_float ? 4 : 8;
    if (f->layout == WFH5_COMPOUND) {
        rc = read_member(f, rn.c_str(), c, base, bs, (hsize_t)row0, (hsize_t)row1, out);
    } else {
        hsize_t n0 = 0, c0 = 0;
        dims2(d, &n0, &c0);
        hid_t sp = H5Dget_space(d);
        const bool two_d = H5Sget_simple_extent_ndims(sp) == 2;
        H5Sclose(sp);
        rc = read_slab(d, (hsize_t)row0, (hsize_t)row1, c, two_d, base, out);
        H5Dclose(d);
    }
    if (rc != WFH5_OK) set_err("HDF5 read of %s failed", rn.c_str());
    return rc;
}
