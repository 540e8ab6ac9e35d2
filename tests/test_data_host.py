"""Host half of the `neuronika-data` mirror (host/data.{hpp,cpp}) against the reference crate's own tests
(neuronika-data/src/test.rs, transcribed into tests/golden/reference_fixtures.json["data"]).  No GPU needed:
without a device the records live in ordinary (not page-locked) host memory."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def nd():
    import neuronika_amd
    return neuronika_amd.tape.data


def arr(lit, shape):
    return np.asarray(lit, np.float32).reshape(shape)


def test_dataset_golden(nd, golden):
    c = golden["data"]["dataset"]; t = c["tests"]
    load = lambda: nd.DataLoader().without_headers().from_string(c["csv"], [10])
    ds = load()
    assert len(ds) == 5 and not ds.is_empty()
    assert np.array_equal(ds.records(), arr(t["from_reader"]["literals"][0], t["from_reader"]["shapes"][0]))
    folds = ds.kfold(2)
    assert len(folds) == 2
    got = [folds[0][0], folds[0][1], folds[1][0], folds[1][1]]
    for g, lit, sh in zip(got, t["kfold"]["literals"], t["kfold"]["shapes"]):
        assert np.array_equal(g.records(), arr(lit, sh))
    for g, lit, sh in zip(ds.batch(3), t["batch"]["literals"], t["batch"]["shapes"]):
        assert np.array_equal(g, arr(lit, sh))
    assert len(ds.batch(3)) == 2 and len(ds.batch(3, True)) == 1
    assert np.array_equal(ds.batch(3, True)[0], arr(t["drop_last"]["literals"][0], t["drop_last"]["shapes"][0]))
    parts = ds.split([1, 1, 1, 2])
    for g, lit, sh in zip(parts, t["split"]["literals"], t["split"]["shapes"]):
        assert np.array_equal(g.records(), arr(lit, sh))
    with pytest.raises(RuntimeError, match="do not cover the whole dataset"):
        ds.split([1, 1])
    before = ds.records().copy()
    ds.shuffle_with_seed(7)
    after = ds.records()
    assert sorted(map(tuple, after)) == sorted(map(tuple, before))          # a permutation of the rows
    ds2 = load(); ds2.shuffle_with_seed(7)
    assert np.array_equal(ds2.records(), after)                             # reproducible for a seed


def test_labeled_dataset_golden(nd, golden):
    c = golden["data"]["labeled_dataset"]; t = c["tests"]
    ds = nd.DataLoader().with_labels(c["label_columns"]).without_headers().from_string(c["csv"], [10], [2])
    lit, sh = t["from_reader"]["literals"], t["from_reader"]["shapes"]
    assert np.array_equal(ds.records(), arr(lit[0], sh[0])) and np.array_equal(ds.labels(), arr(lit[1], sh[1]))
    flat = []
    for train, test in ds.kfold(2):
        flat += [train.records(), train.labels(), test.records(), test.labels()]
    for g, l, s in zip(flat, t["kfold"]["literals"], t["kfold"]["shapes"]):
        assert np.array_equal(g, arr(l, s))
    flat = [a for pair in ds.batch(3) for a in pair]
    for g, l, s in zip(flat, t["batch"]["literals"], t["batch"]["shapes"]):
        assert np.array_equal(g, arr(l, s))
    flat = [a for pair in ds.batch(3, True) for a in pair]
    assert len(flat) == 2
    for g, l, s in zip(flat, t["drop_last"]["literals"], t["drop_last"]["shapes"]):
        assert np.array_equal(g, arr(l, s))
    flat = []
    for part in ds.split([1, 1, 1, 2]):
        flat += [part.records(), part.labels()]
    for g, l, s in zip(flat, t["split"]["literals"], t["split"]["shapes"]):
        assert np.array_equal(g, arr(l, s))
    r0, l0 = ds.records().copy(), ds.labels().copy()
    ds.shuffle_with_seed(3)
    pairs0 = sorted((tuple(a), tuple(b)) for a, b in zip(r0, l0))
    pairs1 = sorted((tuple(a), tuple(b)) for a, b in zip(ds.records(), ds.labels()))
    assert pairs0 == pairs1                                                 # records and labels move together


def test_index_logic_and_errors(nd):
    assert nd.batch_ranges(10, 4, False) == [(0, 4), (4, 4), (8, 2)]
    assert nd.batch_ranges(10, 4, True) == [(0, 4), (4, 4)]
    assert nd.batch_ranges(8, 4, True) == [(0, 4), (4, 4)]                  # equal chunks: nothing dropped
    assert nd.batch_ranges(3, 4, True) == [(0, 3)]                          # a single short chunk is kept (lib.rs:662-674)
    assert nd.batch_ranges(0, 4, False) == []
    folds = nd.kfold_ids(10, 3)                                             # step = 1 + 9 // 3 = 4
    assert [f[1] for f in folds] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]
    assert folds[1][0] == [0, 1, 2, 3, 8, 9]
    p = nd.shuffle_permutation(100, 5)
    assert sorted(p) == list(range(100)) and p != list(range(100)) and p == nd.shuffle_permutation(100, 5)
    with pytest.raises(RuntimeError, match="folds must be"):
        nd.kfold_ids(5, 1)
    with pytest.raises(RuntimeError, match="labels were not provided"):
        nd.DataLoader().with_labels([])
    with pytest.raises(RuntimeError, match="duplicated labels"):
        nd.DataLoader().with_labels([2, 2])
    with pytest.raises(RuntimeError, match="empty records"):
        nd.DataLoader().without_headers().from_string("1,2\n", [0])
    with pytest.raises(RuntimeError, match="ShapeError"):
        nd.DataLoader().without_headers().from_string("1,2,3\n", [2])
    with pytest.raises(RuntimeError, match="not a number"):
        nd.DataLoader().without_headers().from_string("1,x\n", [2])
    hdr = nd.DataLoader().with_delimiter(";").from_string("a;b\n1.5;2\n3;-4e-1\n", [2])     # headers on by default
    assert np.array_equal(hdr.records(), np.array([[1.5, 2], [3, -0.4]], np.float32))
