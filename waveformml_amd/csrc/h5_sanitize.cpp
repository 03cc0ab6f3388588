// h5_sanitize.cpp -- standalone driver of libwfh5's C ABI (include/wfh5.h) for the sanitizer build:
//   make -C waveformml_amd/csrc asan      -> ../lib/h5_sanitize_asan   (h5reader.cpp + this file, -fsanitize=address,undefined)
//   ../lib/h5_sanitize_asan <dir with *.h5 fixtures> [scratch dir]
// h5reader.cpp parses chunk addresses itself, pread()s raw chunks and inflates them outside libhdf5 (the reference
// reads the same files through h5py, src/datasets/HDF5Dataset.py:430-476): this is the code a malformed file reaches
// first.  The driver (no Python: AddressSanitizer cannot be preloaded under this image's interpreter)
//   1. walks every *.h5 under the directory through every entry point -- whole tables, ragged row ranges, every member,
//      event ranges, labels -- under every table name the fixtures use;
//   2. repeats the walk on DAMAGED copies of each file: truncated at several lengths, and with bytes flipped at
//      pseudo-random offsets (fixed seed) INSIDE THE DATASETS' PAYLOAD -- the raw (gzip) chunks and contiguous data
//      blocks, whose file ranges the driver takes from the intact file -- i.e. the bytes this library reads, inflates
//      and converts itself.  Flips in the object headers / B-trees are libhdf5's to survive, and the image's libhdf5
//      1.10.6 does not (H5O_attr_shared_decode copies 8 MB out of a 128-byte block on one flipped header bit: found by
//      the first version of this driver, not reachable from our code path on well-formed metadata).  A damaged file may
//      fail (WFH5_EIO / WFH5_EFORMAT / WFH5_EINVAL) or read garbage VALUES; it must never crash, overrun a buffer or
//      trip the sanitizers.
// Exit code 0 = every call returned, nothing was reported; the sanitizers abort the process otherwise.
#include <dirent.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <vector>

#include <hdf5.h>

#include "../../include/wfh5.h"

static const char *TABLES[] = {"WaveformPairs", "Waveform3DPairs", "WaveformNorm", "EventLabels"};
static const char *NAMES[][2] = {{"coord", "waveform"}, {"det", "pulse"}, {"", ""}};
static long g_calls = 0, g_ok = 0, g_failed = 0;

static void note(int rc) {
    ++g_calls;
    if (rc == WFH5_OK) ++g_ok; else ++g_failed;
}

static void list_h5(const std::string &dir, std::vector<std::string> *out) {
    DIR *d = opendir(dir.c_str());
    if (!d) return;
    while (dirent *e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        const std::string p = dir + "/" + e->d_name;
        struct stat st;
        if (stat(p.c_str(), &st) != 0) continue;
        if (S_ISDIR(st.st_mode)) list_h5(p, out);
        else if (p.size() > 3 && p.compare(p.size() - 3, 3, ".h5") == 0) out->push_back(p);
    }
    closedir(d);
}

// every entry point on one open table; `trust` = the file is intact (results are checked for plausibility)
static void walk(wfh5_file *f, bool named_empty) {
    wfh5_info info;
    memset(&info, 0, sizeof(info));
    int rc = wfh5_get_info(f, &info);
    note(rc);
    if (rc != WFH5_OK) return;
    // a damaged header may announce absurd sizes: the reader must refuse or stay inside what it allocates; the driver
    // bounds ITS buffers (a row range of at most 4096 rows at a time)
    const int64_t n = info.n_rows;
    if (!named_empty && n > 0 && info.coord_cols > 0 && info.coord_cols <= 8 && info.feat_cols > 0 && info.feat_cols < (1 << 16)) {
        const int64_t spans[][2] = {{0, n}, {0, 1}, {n / 3, n / 3 + 7}, {n - 1, n}, {n / 2, n}, {n, n}, {0, n + 5}, {-1, 3}, {5, 2}};
        for (const auto &sp : spans) {
            int64_t r0 = sp[0], r1 = sp[1];
            int64_t want = r1 > r0 ? r1 - r0 : 0;
            if (want > 4096) { r1 = r0 + 4096; want = 4096; }
            std::vector<int32_t> c((size_t)(want > 0 ? want : 1) * info.coord_cols);
            std::vector<float> x((size_t)(want > 0 ? want : 1) * info.feat_cols);
            note(wfh5_read_rows(f, r0, r1, c.data(), x.data(), 1.0f / 16383.0f));
            note(wfh5_read_rows(f, r0, r1, nullptr, x.data(), 1.0f));
            note(wfh5_read_rows(f, r0, r1, c.data(), nullptr, 1.0f));
        }
        if (info.n_events > 0 && info.n_events < (1 << 20)) {
            const int64_t ev[][2] = {{0, info.n_events - 1}, {0, 0}, {info.n_events / 2, info.n_events - 1}, {info.n_events, info.n_events + 1}, {-1, 0}, {3, 1}};
            for (const auto &e : ev)
                for (int col = 0; col <= info.coord_cols; ++col) {
                    int64_t a = -7, b = -7;
                    note(wfh5_event_rows(f, col < info.coord_cols ? col : info.coord_cols - 1, e[0], e[1], &a, &b));
                }
        }
    }
    if (info.n_labels > 0 && info.n_labels < (1 << 20)) {
        std::vector<int64_t> y((size_t)info.n_labels + 1);
        note(wfh5_read_labels(f, 0, info.n_labels, y.data()));
        note(wfh5_read_labels(f, info.n_labels / 2, info.n_labels, y.data()));
        note(wfh5_read_labels(f, 0, info.n_labels + 1, y.data()));
    }
    const char *members[] = {nullptr, "", "coord", "waveform", "PID", "PID8", "PID16", "PID64", "phys", "EZ", "labels", "label", "weight",
                             "evt", "det", "pulse", "nope"};
    for (const char *m : members) {
        int64_t rows = 0;
        int32_t cols = 0, fl = 0, es = 0;
        rc = wfh5_member_info(f, m, &rows, &cols, &fl, &es);
        note(rc);
        if (rc != WFH5_OK || rows <= 0 || cols <= 0 || cols > (1 << 16)) continue;
        const int64_t take = rows < 2048 ? rows : 2048;
        std::vector<int64_t> buf((size_t)take * cols + 8);           // 8 bytes per element covers both widenings
        note(wfh5_read_member(f, m, 0, take, 0, buf.data()));
        note(wfh5_read_member(f, m, 0, take, 1, buf.data()));
        note(wfh5_read_member(f, m, rows - 1, rows, fl, buf.data()));
        note(wfh5_read_member(f, m, rows, rows + 1, fl, buf.data()));
        note(wfh5_read_member(f, m, 2, 1, fl, buf.data()));
    }
}

static void exercise(const std::string &path) {
    for (const char *table : TABLES) {
        wfh5_file *f = nullptr;
        int rc = wfh5_open(path.c_str(), table, &f);
        note(rc);
        if (rc == WFH5_OK && f) {
            walk(f, false);
            wfh5_close(f);
        }
        for (const auto &nm : NAMES) {
            f = nullptr;
            rc = wfh5_open_named(path.c_str(), table, nm[0], nm[1], &f);
            note(rc);
            if (rc == WFH5_OK && f) {
                walk(f, nm[0][0] == 0);
                wfh5_close(f);
            }
        }
    }
    wfh5_file *f = nullptr;
    note(wfh5_open(path.c_str(), "no_such_table", &f));
    if (f) wfh5_close(f);
}

// file ranges [offset, offset + bytes) of the payload of every dataset the reader touches, from the INTACT file
static void dataset_ranges(hid_t d, std::vector<std::pair<size_t, size_t>> *out) {
    hid_t plist = H5Dget_create_plist(d);
    if (plist >= 0 && H5Pget_layout(plist) == H5D_CHUNKED) {
        hid_t space = H5Dget_space(d);
        hsize_t n = 0;
        if (space >= 0 && H5Dget_num_chunks(d, space, &n) >= 0)
            for (hsize_t i = 0; i < n; ++i) {
                hsize_t off[8];
                unsigned mask = 0;
                haddr_t addr = 0;
                hsize_t size = 0;
                if (H5Dget_chunk_info(d, space, i, off, &mask, &addr, &size) >= 0 && addr != HADDR_UNDEF && size > 0)
                    out->push_back({(size_t)addr, (size_t)size});
            }
        if (space >= 0) H5Sclose(space);
    } else {
        const haddr_t addr = H5Dget_offset(d);
        const hsize_t size = H5Dget_storage_size(d);
        if (addr != HADDR_UNDEF && size > 0) out->push_back({(size_t)addr, (size_t)size});
    }
    if (plist >= 0) H5Pclose(plist);
}

static void payload_ranges(const std::string &path, std::vector<std::pair<size_t, size_t>> *out) {
    H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
    hid_t f = H5Fopen(path.c_str(), H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f < 0) return;
    for (const char *table : TABLES) {
        if (H5Lexists(f, table, H5P_DEFAULT) <= 0) continue;
        hid_t o = H5Oopen(f, table, H5P_DEFAULT);
        if (o < 0) continue;
        H5I_type_t t = H5Iget_type(o);
        if (t == H5I_DATASET) {
            dataset_ranges(o, out);
        } else if (t == H5I_GROUP) {
            for (const char *m : {"coord", "waveform", "labels"}) {
                if (H5Lexists(o, m, H5P_DEFAULT) <= 0) continue;
                hid_t d = H5Dopen2(o, m, H5P_DEFAULT);
                if (d >= 0) {
                    dataset_ranges(d, out);
                    H5Dclose(d);
                }
            }
        }
        H5Oclose(o);
    }
    H5Fclose(f);
}

static bool read_all(const std::string &path, std::vector<unsigned char> *out) {
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END);
    const long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    out->resize((size_t)(n > 0 ? n : 0));
    const size_t got = n > 0 ? fread(out->data(), 1, (size_t)n, fp) : 0;
    fclose(fp);
    return got == out->size();
}

static bool write_all(const std::string &path, const unsigned char *p, size_t n) {
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) return false;
    const size_t put = n ? fwrite(p, 1, n, fp) : 0;
    fclose(fp);
    return put == n;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s <fixture dir> [scratch dir]\n", argv[0]);
        return 2;
    }
    const std::string scratch = argc > 2 ? argv[2] : "/tmp";
    wfh5_set_threads(2);
    std::vector<std::string> files;
    list_h5(argv[1], &files);
    if (files.empty()) {
        fprintf(stderr, "no *.h5 under %s\n", argv[1]);
        return 2;
    }
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() {
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        return (uint32_t)(lcg >> 33);
    };
    long damaged = 0, flipped = 0;
    for (const std::string &p : files) {
        exercise(p);                                              // the intact file
        std::vector<unsigned char> bytes;
        if (!read_all(p, &bytes) || bytes.size() < 64) continue;
        const std::string tmp = scratch + "/wfh5_sanitize_" + std::to_string((long)getpid()) + ".h5";
        const size_t cuts[] = {bytes.size() - 1, bytes.size() * 3 / 4, bytes.size() / 2, 2048, 9};
        for (size_t cut : cuts) {                                 // truncated copies
            if (cut >= bytes.size() || !write_all(tmp, bytes.data(), cut)) continue;
            exercise(tmp);
            ++damaged;
        }
        std::vector<std::pair<size_t, size_t>> ranges;
        payload_ranges(p, &ranges);
        size_t payload = 0;
        for (const auto &r : ranges) payload += r.second;
        for (int round = 0; round < 10 && payload > 0; ++round) { // copies with flipped payload bytes
            std::vector<unsigned char> b = bytes;
            const int flips = 1 + (int)(rnd() % 8);
            for (int i = 0; i < flips; ++i) {
                size_t k = rnd() % payload, at = 0;               // the k-th payload byte of the file
                for (const auto &r : ranges) {
                    if (k < r.second) {
                        at = r.first + k;
                        break;
                    }
                    k -= r.second;
                }
                if (at >= b.size()) continue;
                // early bytes of a chunk are the deflate stream's header / first block: flip those more often
                if (rnd() % 3 == 0)
                    for (const auto &r : ranges)
                        if (at >= r.first && at < r.first + r.second) {
                            at = r.first + rnd() % (r.second < 16 ? r.second : 16);
                            break;
                        }
                b[at] ^= (unsigned char)(1u << (rnd() % 8));
                if (rnd() % 4 == 0) b[at] = (unsigned char)rnd();
                ++flipped;
            }
            if (!write_all(tmp, b.data(), b.size())) continue;
            exercise(tmp);
            ++damaged;
        }
        unlink(tmp.c_str());
    }
    printf("h5_sanitize: %zu files, %ld damaged copies (%ld payload bytes flipped), %ld calls (%ld ok, %ld refused), last error: %s\n",
           files.size(), damaged, flipped, g_calls, g_ok, g_failed, wfh5_last_error());
    return 0;
}
