"""Native HDF5 -> COO reader (include/wfh5.h, waveformml_amd/psd/h5data.py) against the committed fixtures
tests/golden/h5/ (written by tests/golden/make_h5_fixtures.py with h5py; expected arrays in expected.npz).

Behaviour under test is the reference's HDF5Dataset (src/datasets/HDF5Dataset.py:152-217, 225-347): file ordering
and per-directory event budget, event-range slicing, directory-index / dataset labels, 1/(2^14-1) normalisation, and
the collate of src/engineering/PSDDataModule.py:10-20 over the items.  CPU only: the reader is host code.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from waveformml_amd.psd import data, h5data

H5 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")
EXP = np.load(os.path.join(H5, "expected.npz"))


def _slice_events(coords, col, e0, e1):
    """numpy restatement of reference HDF5Dataset.py:238-248 (first row of e0, first row of e1 + 1)."""
    a = 0 if e0 == 0 else int(np.where(coords[:, col] == e0)[0][0])
    hit = np.where(coords[:, col] == e1 + 1)[0]
    return a, (int(hit[0]) if len(hit) else len(coords))


@pytest.mark.parametrize("rel,table,layout,cols,is_float", [
    ("Gamma/a_WaveformPairSim.h5", "WaveformPairs", h5data.WFH5_COMPOUND, 3, False),
    ("Gamma/a_Waveform3DPairSim.h5", "Waveform3DPairs", h5data.WFH5_COMPOUND, 4, True),
    ("combined/Combined_0_WaveformPairSim.h5", "WaveformPairs", h5data.WFH5_GROUP, 3, False),
])
def test_whole_table_matches_h5py(rel, table, layout, cols, is_float):
    with h5data.H5Table(os.path.join(H5, rel), table) as t:
        c_ref, w_ref = EXP[rel + "/coord"], EXP[rel + "/waveform"]
        assert (t.layout, t.coord_cols, t.feat_is_float) == (layout, cols, is_float)
        assert (t.n_rows, t.feat_cols) == (len(c_ref), w_ref.shape[1])
        assert t.n_events == c_ref[:, -1].max() + 1
        c, f = t.read_rows(0, t.n_rows)
        assert c.dtype == torch.int32 and f.dtype == torch.float32
        assert np.array_equal(c.numpy(), c_ref)
        assert np.array_equal(f.numpy(), w_ref.astype(np.float32))       # int16 -> float32 is exact
        # a scaled partial read equals the reference's `vals *= MAX_RANGE_INV` on the float32 array
        c2, f2 = t.read_rows(3, 11, h5data.MAX_RANGE_INV)
        assert np.array_equal(c2.numpy(), c_ref[3:11])
        want = w_ref[3:11].astype(np.float32)
        want *= h5data.MAX_RANGE_INV
        assert np.array_equal(f2.numpy(), want)


def test_event_rows_follow_first_occurrence_rule():
    rel = "Electron/a_WaveformPairSim.h5"
    c_ref = EXP[rel + "/coord"]
    with h5data.H5Table(os.path.join(H5, rel), "WaveformPairs") as t:
        for e0, e1 in ((0, 0), (0, 4), (3, 7), (5, 11), (11, 11)):
            assert t.event_rows(e0, e1, 2) == _slice_events(c_ref, 2, e0, e1)
        with pytest.raises(h5data.H5Error):
            t.event_rows(40, 41, 2)
        with pytest.raises(h5data.H5Error):
            t.read_rows(0, t.n_rows + 1)


def test_labels_dataset_is_widened_to_int64():
    rel = "combined/Combined_0_WaveformPairSim.h5"
    with h5data.H5Table(os.path.join(H5, rel), "WaveformPairs") as t:
        assert t.n_labels == 11
        y = t.read_labels(2, 9)
        assert y.dtype == torch.int64 and np.array_equal(y.numpy(), EXP[rel + "/labels"][2:9].astype(np.int64))
    with h5data.H5Table(os.path.join(H5, "Gamma/a_WaveformPairSim.h5"), "WaveformPairs") as t:
        assert t.n_labels == 0
        with pytest.raises(h5data.H5Error):
            t.read_labels(0, 1)


def test_open_errors_are_reported_not_crashed():
    with pytest.raises(h5data.H5Error, match="cannot open"):
        h5data.H5Table(os.path.join(H5, "nope.h5"), "WaveformPairs")
    with pytest.raises(h5data.H5Error, match="no object named"):
        h5data.H5Table(os.path.join(H5, "Gamma/a_WaveformPairSim.h5"), "Waveform3DPairs")
    with pytest.raises(h5data.H5Error):                       # not an HDF5 file at all
        h5data.H5Table(os.path.join(H5, "expected.npz"), "WaveformPairs")


def test_dataset_orders_files_round_robin_and_applies_the_event_budget():
    ds = h5data.PulseDataset2D([os.path.join(H5, "Gamma"), os.path.join(H5, "Electron")], 10, normalize=True)
    got = [(os.path.relpath(d["file_path"], H5), d["event_range"], d["dir_index"], d["n_events"])
           for d in ds.info["data_info"]]
    # Gamma/a (9 ev) -> Electron/a (12 ev, cut to 10) -> Gamma/b (5 ev, 1 left of the budget)
    assert got == [("Gamma/a_WaveformPairSim.h5", [0, 8], 0, 9), ("Electron/a_WaveformPairSim.h5", [0, 9], 1, 12),
                   ("Gamma/b_WaveformPairSim.h5", [0, 0], 0, 5)]
    assert ds.n_events == [10, 10] and len(ds) == 3
    for i, (rel, (e0, e1), dir_index, _) in enumerate(got):
        (c, f), y = ds[i]
        c_ref, w_ref = EXP[rel + "/coord"], EXP[rel + "/waveform"]
        a, b = _slice_events(c_ref, 2, e0, e1)
        want = w_ref[a:b].astype(np.float32)
        want *= h5data.MAX_RANGE_INV
        assert np.array_equal(c.numpy(), c_ref[a:b]) and np.array_equal(f.numpy(), want)
        assert y.dtype == torch.int64 and y.tolist() == [dir_index] * (e1 + 1 - e0)
        assert f.max() <= 1.0


def test_single_directory_budget_and_excludes():
    gamma = os.path.join(H5, "Gamma")
    ds = h5data.PulseDataset2D([gamma], 11)
    assert [d["event_range"] for d in ds.info["data_info"]] == [[0, 8], [0, 1]]
    ds = h5data.PulseDataset2D([gamma], 100, file_excludes=[os.path.join(gamma, "a_WaveformPairSim.h5")])
    assert [os.path.basename(p) for p in ds.ordered_file_set] == ["b_WaveformPairSim.h5"]
    with pytest.raises(RuntimeError, match="No hdf5 datasets found"):
        h5data.PulseDataset3D([os.path.join(H5, "combined")], 10)
    with pytest.raises(RuntimeError, match="not a valid directory"):
        h5data.PulseDataset2D([os.path.join(H5, "Proton")], 10)


def test_3d_items_slice_on_the_batch_column_and_collate():
    rel = "Gamma/a_Waveform3DPairSim.h5"
    c_ref, w_ref = EXP[rel + "/coord"], EXP[rel + "/waveform"]
    ds = h5data.PulseDataset3D([os.path.join(H5, "Gamma")], 5, use_half=True)
    (c, f), y = ds[0]
    a, b = _slice_events(c_ref, 3, 0, 4)
    assert np.array_equal(c.numpy(), c_ref[a:b]) and f.dtype == torch.float16 and len(y) == 5
    assert torch.equal(f, torch.from_numpy(w_ref[a:b]).half())
    # two items through the reference collate: the second item's event ids continue after the first's
    full = h5data.PulseDataset3D([os.path.join(H5, "Gamma")], 7)
    (coords, feats), labels = data.collate_fn_3d([full[0], ds[0]])
    assert coords.shape[0] == len(c_ref) + (b - a) and labels.shape[0] == 12
    assert coords[len(c_ref):, 3].min() == 7 and coords[:, 3].max() == 11
    assert torch.equal(feats[:len(c_ref)].float(), torch.from_numpy(w_ref))


def test_group_layout_with_labels_dataset_and_label_map():
    rel = "combined/Combined_0_WaveformPairSim.h5"
    c_ref, l_ref = EXP[rel + "/coord"], EXP[rel + "/labels"].astype(np.int64)
    args = ([os.path.join(H5, "combined")], "*WaveformPairSim.h5", "WaveformPairs", "coord", "waveform", 8)
    ds = h5data.HDF5Dataset(*args, label_name="labels")
    (c, f), y = ds[0]
    a, b = _slice_events(c_ref, 2, 0, 7)
    assert np.array_equal(c.numpy(), c_ref[a:b]) and np.array_equal(y.numpy(), l_ref[:8])
    ds = h5data.HDF5Dataset(*args, label_name="labels", label_map={"2": 0})
    assert np.array_equal(ds[0][1].numpy(), np.where(l_ref[:8] == 2, 0, l_ref[:8]))


def test_dataloader_workers_read_through_their_own_handles():
    ds = h5data.PulseDataset2D([os.path.join(H5, "Gamma"), os.path.join(H5, "Electron")], 10)
    loader = torch.utils.data.DataLoader(ds, batch_size=3, num_workers=2, collate_fn=data.collate_fn)
    (coords, feats), labels = next(iter(loader))
    assert labels.tolist() == [0] * 9 + [1] * 10 + [0]
    assert coords[:, 2].max() == 19 and feats.shape[0] == coords.shape[0]


def test_gzip_chunked_compound_table_takes_the_parallel_path_and_bisects_events():
    """Chunked + gzip compound table (64-record chunks): bulk reads inflate the raw chunks on worker threads, the event
    search bisects a sorted file; both must give exactly what h5py read."""
    rel = "gz/sorted_Waveform3DPairSim.h5"
    c_ref, w_ref = EXP[rel + "/coord"], EXP[rel + "/waveform"]
    for threads in (1, 3):
        h5data.set_threads(threads)
        with h5data.H5Table(os.path.join(H5, rel), "Waveform3DPairs") as t:
            assert (t.n_rows, t.n_events, t.coord_cols, t.feat_cols) == (len(c_ref), 23, 4, 2)
            for r0, r1 in ((0, t.n_rows), (1, 65), (63, 129), (100, 500), (t.n_rows - 70, t.n_rows), (5, 6)):
                c, f = t.read_rows(r0, r1, 0.5)
                assert np.array_equal(c.numpy(), c_ref[r0:r1]) and np.array_equal(f.numpy(), w_ref[r0:r1] * np.float32(0.5))
            for e0, e1 in ((0, 22), (0, 0), (4, 9), (22, 22), (11, 21)):
                assert t.event_rows(e0, e1, 3) == _slice_events(c_ref, 3, e0, e1)
            with pytest.raises(h5data.H5Error):
                t.event_rows(23, 25, 3)
    h5data.set_threads(4)


def test_unsorted_event_ids_fall_back_to_the_first_occurrence_scan():
    rel = "gz/unsorted_Waveform3DPairSim.h5"
    c_ref = EXP[rel + "/coord"]
    assert not np.all(np.diff(c_ref[:, 3]) >= 0)
    with h5data.H5Table(os.path.join(H5, rel), "Waveform3DPairs") as t:
        for e0, e1 in ((2, 3), (4, 7), (1, 2)):                # e0 and e1 + 1 both occur: the reference's rule applies
            assert t.event_rows(e0, e1, 3) == _slice_events(c_ref, 3, e0, e1)


def _dm_config(dataset_class, **extra):
    from waveformml_amd.psd.config import DictionaryUtility
    cfg = {"system_config": {"half_precision": 0},
           "dataset_config": {"imports": ["waveformml_amd.psd.PulseDataset"], "dataset_class": dataset_class,
                              "base_path": H5, "paths": ["Gamma", "Electron"], "n_train": 9, "n_validate": 3, "n_test": 3,
                              "dataset_params": {"data_cache_size": 1},
                              "dataloader_params": {"batch_size": 2, "num_workers": 0}}}
    cfg["dataset_config"].update(extra.pop("dataset_config", {}))
    cfg["system_config"].update(extra.pop("system_config", {}))
    return DictionaryUtility.to_object(cfg)


def test_data_module_builds_splits_from_the_config_like_the_reference():
    """PSDDataModule(config, device) as main.py uses it (reference src/engineering/PSDDataModule.py:22-151): the dataset
    class comes from dataset_config.{imports, dataset_class} and is called (config, split, n, device, **dataset_params);
    directories = base_path + paths (label = directory index), features normalised; validation excludes the training
    files, test excludes both; loaders batch ITEMS (one item = one file's event range) through the reference's collate."""
    from waveformml_amd.psd.PSDDataModule import PSDDataModule
    dm = PSDDataModule(_dm_config("PulseDataset.PulseDataset2D"), "cpu")
    assert dm.ntype == 2 and dm.total_train == 18 and dm.half_precision == 0
    dm.setup("fit")
    train_files = [os.path.relpath(p, H5) for p in dm.train_dataset.get_file_list()]
    assert train_files == ["Gamma/a_WaveformPairSim.h5", "Electron/a_WaveformPairSim.h5"]
    assert dm.train_dataset.n_events == [9, 9] and dm.train_dataset.n_categories == 2
    assert [os.path.relpath(p, H5) for p in dm.train_excludes] == train_files
    # validation: Gamma/b is left, but the Electron directory has no file the training set did not take -> the
    # reference's error (src/datasets/HDF5Dataset.py:150-152)
    with pytest.raises(RuntimeError, match="No remaining datasets available"):
        dm.setup("test")
    loader = dm.train_dataloader()
    batches = list(loader)
    assert len(batches) == 1
    (coords, feats), labels = batches[0]
    assert coords.shape[1] == 3 and feats.dtype == torch.float32 and float(feats.max()) <= 1.0
    assert labels.shape[0] == 18 and sorted(labels.tolist()) == [0] * 9 + [1] * 9
    assert int(coords[:, 2].max()) == 17 and coords[:, 2].unique().numel() <= 18     # event ids continue across items


def test_data_module_split_excludes_and_half_precision():
    from waveformml_amd.psd.PSDDataModule import PSDDataModule
    # one directory with two files: train takes a (7 events), validation must come from b, nothing is left for test
    cfg = _dm_config("PulseDataset.PulseDataset3D", dataset_config={"paths": ["Gamma"], "n_train": 7, "n_validate": 5},
                     system_config={"half_precision": 1})
    dm = PSDDataModule(cfg, "cpu")
    assert cfg.dataset_config.dataset_params.use_half is True          # propagated as the reference does (:29-33)
    with pytest.raises(RuntimeError, match="No remaining datasets available"):
        dm.setup("test")
    assert [os.path.basename(p) for p in dm.val_dataset.get_file_list()] == ["b_Waveform3DPairSim.h5"]
    (coords, feats), labels = next(iter(dm.val_dataloader()))
    assert coords.shape[1] == 4 and feats.dtype == torch.float16 and labels.tolist() == [0] * 4
    assert int(coords[:, 3].max()) == 3          # b holds 4 events, fewer than the budget of 5
    for key in ("train_config", "data_prep"):
        bad = _dm_config("PulseDataset.PulseDataset2D", dataset_config={key: "shuffle"})
        with pytest.raises(NotImplementedError):
            PSDDataModule(bad, "cpu")


def test_packed_loader_hands_over_the_same_batches():
    """psd/data.PackedLoader (one shared-memory buffer per batch between worker and trainer process) against the plain
    DataLoader with the reference's collate: identical tensors, dtypes and shapes, for both layouts."""
    from torch.utils.data import DataLoader
    from waveformml_amd.psd import data
    for layout, fn in (("3d", data.collate_fn_3d), ("2d", data.collate_fn)):
        ds = data.SyntheticPulseDataset(5, 4, 24, layout=layout)
        plain = list(DataLoader(ds, batch_size=2, collate_fn=fn))
        for group in (1, 2, 4):                     # batches per worker -> trainer message
            loader = data.PackedLoader(ds, fn, group=group, batch_size=2, num_workers=2)
            packed = list(loader)
            assert len(plain) == len(packed) == len(loader) == 3
            for a, b in zip(plain, packed):
                for x, y in ((a[0][0], b[0][0]), (a[0][1], b[0][1]), (a[1], b[1])):
                    assert x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y)


def test_packed_loader_exposes_its_sampler_for_per_epoch_reshuffling():
    """The Trainer calls ``loader.sampler.set_epoch(epoch)`` on whatever loader it is given (rank sharding under
    torch.distributed); a PackedLoader hands its DataLoader's sampler through, and two ranks' shares stay disjoint."""
    from torch.utils.data.distributed import DistributedSampler
    from waveformml_amd.psd import data
    ds = data.SyntheticPulseDataset(8, 2, 16, layout="3d")
    seen = []
    for rank in (0, 1):
        sampler = DistributedSampler(ds, num_replicas=2, rank=rank, shuffle=True, seed=3)
        loader = data.PackedLoader(ds, data.collate_fn_3d, batch_size=1, num_workers=1, sampler=sampler)
        assert loader.sampler is sampler and len(loader) == 4
        sampler.set_epoch(0)
        e0 = list(iter(sampler))
        sampler.set_epoch(1)
        e1 = list(iter(sampler))
        assert sorted(e0) != sorted(range(8)) and e0 != e1            # a share, reshuffled per epoch
        assert sum(1 for _ in loader) == 4
        seen.append(set(e1))
    assert seen[0].isdisjoint(seen[1]) and seen[0] | seen[1] == set(range(8))


# ---- round 3: the sibling tasks' files (reference src/datasets/HDF5Dataset.py:186-217, 250-347, 404-427) ------------
R3 = os.path.join(H5, "r3")
EXP3 = np.load(os.path.join(os.path.dirname(H5), "expected_r3.npz"))
PID_MAP = {"1": 0, "4": 1, "6": 2, "256": 3, "258": 2, "512": 4}          # config/examples/IoniClassifierCNN.json:67-74


def _ioni(**kw):
    return h5data.HDF5Dataset([os.path.join(R3, "ioni")], "*WaveformPairSim.h5", "WaveformPairs", "coord", "waveform", 12, **kw)


def test_per_row_label_column_by_member_name_through_the_label_map():
    """`label_name: "PID"` (IoniClassifierCNN.json:75-76): one label per ROW of the compound table, mapped to class
    indices in place, int64 -- the item of a partly used file is cut at the event boundary like the features."""
    ds = _ioni(label_name="PID", label_map=PID_MAP)
    assert len(ds) == 2                                     # 9 events of run_1, 3 of run_2's 7 (budget 12)
    lut = {int(k): v for k, v in PID_MAP.items()}
    (c, f), y = ds[0]
    want = np.vectorize(lut.get)(EXP3["ioni/run_1/PID"])
    assert y.dtype == torch.int64 and np.array_equal(y.numpy(), want) and len(y) == len(c)
    assert np.array_equal(c.numpy(), EXP3["ioni/run_1/coord"])
    assert np.array_equal(f.numpy(), EXP3["ioni/run_1/waveform"].astype(np.float32))
    (c, f), y = ds[1]
    n = int(np.searchsorted(EXP3["ioni/run_2/coord"][:, 2], 3))          # rows of events 0..2
    assert len(c) == n and np.array_equal(y.numpy(), np.vectorize(lut.get)(EXP3["ioni/run_2/PID"][:n]))
    # a float member as the label: regression targets, float32, [n, len] (SegQuantifier.json:70 "phys")
    (c, f), y = _ioni(label_name="phys")[0]
    assert y.dtype == torch.float32 and np.array_equal(y.numpy(), EXP3["ioni/run_1/phys"])
    (c, f), y = _ioni(label_name="EZ", normalize=True)[0]
    assert np.array_equal(y.numpy(), EXP3["ioni/run_1/EZ"])
    np.testing.assert_allclose(f.numpy(), EXP3["ioni/run_1/waveform"].astype(np.float32) / (2 ** 14 - 1), rtol=1e-6)


def test_additional_fields_come_back_as_the_reference_list():
    """`additional_fields: ["phys"]` (IoniClassifierCNN.json:88-89): vals = [feats, *fields], the fields in their stored
    type, every entry cut to the item's rows."""
    ds = _ioni(label_name="PID", label_map=PID_MAP, additional_fields=["phys", "PID"], use_half=True)
    (c, vals), y = ds[1]
    n = len(c)
    assert isinstance(vals, list) and len(vals) == 3 and vals[0].dtype == torch.float16
    assert vals[1].dtype == torch.float32 and np.array_equal(vals[1].numpy(), EXP3["ioni/run_2/phys"][:n])
    assert vals[2].dtype == torch.int32 and np.array_equal(vals[2].numpy(), EXP3["ioni/run_2/PID"][:n])      # unmapped


def test_label_file_pattern_reads_the_first_member_per_event():
    """`label_file_pattern` (reference :404-427, :483): labels live in the sibling file named by the pattern swap, in
    the `label_name` table, first member, one entry per EVENT of the item's range."""
    ds = _ioni(label_name="EventLabels", label_file_pattern="*Label.h5", label_map={"3": 0})
    (c, f), y = ds[0]
    want = EXP3["ioni/run_1/label"].astype(np.int64)
    want[want == 3] = 0
    assert y.dtype == torch.int64 and np.array_equal(y.numpy(), want)
    (c, f), y = ds[1]
    want = EXP3["ioni/run_2/label"][:3].astype(np.int64)
    want[want == 3] = 0
    assert np.array_equal(y.numpy(), want)
    with pytest.raises(RuntimeError, match="No corresponding label file"):
        _ioni(label_name="EventLabels", label_file_pattern="*Nothing.h5")[0]


def test_other_member_names_and_length_based_ranges():
    """"det" / "pulse" tables without an `nevents` attribute (reference PulseDatasetWaveformNorm, `event_based=False`):
    an item's range counts ROWS."""
    ds = h5data.HDF5Dataset([os.path.join(R3, "pulses")], "*PulseNorm.h5", "WaveformNorm", "det", "pulse", 17,
                            label_name="phys", event_based=False, additional_fields=["PID"])
    assert len(ds) == 1 and ds.info["data_info"][0]["event_range"] == [0, 16] and ds.info["data_info"][0]["n_events"] == 23
    (c, vals), y = ds[0]
    assert c.shape == (17, 1) and np.array_equal(c.numpy()[:, 0], EXP3["pulses/p_1/det"][:17])
    assert np.array_equal(vals[0].numpy(), EXP3["pulses/p_1/pulse"][:17]) and np.array_equal(vals[1].numpy(), EXP3["pulses/p_1/PID"][:17])
    assert y.dtype == torch.float32 and np.array_equal(y.numpy(), EXP3["pulses/p_1/phys"][:17])


def test_member_reads_through_the_c_abi():
    with h5data.H5Table(os.path.join(R3, "ioni", "run_1_WaveformPairSim.h5"), "WaveformPairs") as t:
        rows, cols, fl, es = t.member_info("phys")
        assert (rows, cols, fl, es) == (len(EXP3["ioni/run_1/PID"]), 8, True, 4)
        assert t.member_info("PID")[1:] == (1, False, 4) and t.member_info(None)[1:] == (1, False, 8)        # first member: evt
        assert np.array_equal(t.read_member("PID", 2, 7).numpy(), EXP3["ioni/run_1/PID"][2:7])
        with pytest.raises(h5data.H5Error):
            t.read_member("nope", 0, 1)
        with pytest.raises(h5data.H5Error):
            t.read_member("PID", 0, rows + 1)
    with h5data.H5Table(os.path.join(R3, "ioni", "run_1_Label.h5"), "EventLabels", "", "") as t:
        assert t.n_events == 9 and np.array_equal(t.read_member(None, 0, 9).numpy(), EXP3["ioni/run_1/label"])


# ---- round 4: the reference's label rules per path (src/datasets/HDF5Dataset.py:319-341, :582-585) ---------
R4 = os.path.join(H5, "r4", "labels")
EXP4 = np.load(os.path.join(os.path.dirname(H5), "expected_r4.npz"))


def _r4(**kw):
    return h5data.HDF5Dataset([R4], "*WaveformPairSim.h5", "WaveformPairs", "coord", "waveform", 6, **kw)


def _chained(values, mapping):
    """The reference's convert_label: ``for key, val in label_map.items(): y[y == key] = val`` -- in place, in order."""
    y = np.array(values, copy=True)
    for k, v in mapping.items():
        y[y == (float(k) if y.dtype.kind == "f" else int(k))] = v
    return y


def test_per_row_labels_follow_the_stored_type_and_the_map_is_chained():
    """One label per row: int64 class indices only for a member STORED as int32, float32 for every other stored type
    (int8 / int16 / int64 included) -- reference :331-341; the label map is applied key by key in place, so 1 -> 4 -> 6
    chains (reference :582-585)."""
    m = {"1": 4, "4": 6}                                   # a 1 becomes a 4, which the next key turns into a 6
    want = _chained(EXP4["pid"], m)
    assert set(np.unique(want)) <= {6, 7}
    (c, f), y = _r4(label_name="PID", label_map=m)[0]
    assert y.dtype == torch.int64 and np.array_equal(y.numpy(), want) and np.array_equal(c.numpy(), EXP4["coord"])
    for name in ("PID16", "PID8", "PID64"):
        (c, f), y = _r4(label_name=name, label_map=m)[0]
        assert y.dtype == torch.float32, name
        assert np.array_equal(y.numpy(), want.astype(np.float32)), name


def test_event_level_labels_are_always_int64_also_from_a_float_label_file():
    """Label files (one label per event): the reference maps in place on the stored (float) values and then casts to
    int64 whatever the stored type (:319-327): 2.5 -> 2."""
    ds = _r4(label_name="EventLabels", label_file_pattern="*FLabel.h5", label_map={"3": 1, "1": 0})
    (c, f), y = ds[0]
    want = _chained(EXP4["flabel"], {"3": 1, "1": 0}).astype(np.int64)      # 3.0 -> 1.0 -> 0.0 (chained), 2.5 -> 2
    assert y.dtype == torch.int64 and np.array_equal(y.numpy(), want)
    assert 2 in want or 2.5 not in EXP4["flabel"]
