"""The step in front of the hot path: label files -> label tensors, batch flips with their boxes.

CPU tests pin the oracle's restatement by hand-computed answers AND by golden vectors written by the reference's own functions
(tests/golden/make_golden_data.py imports yogo/data/yogo_dataset.py:24-133 and data_transforms.py:51-98 on stubs of their
torchvision / ruamel imports -> tests/golden/data_step.npz), and cover the host-side parser; GPU tests compare the HIP kernels
with the oracle and with those vectors bit for bit."""

import numpy as np
import pytest
import torch

import yogo_oracle as O


# ---------------------------------------------------------------------------------------------------------------------------
# CPU: oracle known answers + host parsing
# ---------------------------------------------------------------------------------------------------------------------------
def test_oracle_rasteriser_known_answers():
    Sx, Sy = 4, 3
    rows = torch.tensor([[2.0, 0.30, 0.40, 0.45, 0.60],     # centre (0.375, 0.5) -> cell i = floor(0.75 * 4 / 2) = 1, j = 1
                         [5.0, 0.80, 0.05, 0.95, 0.25],     # centre (0.875, 0.15) -> i = 3, j = 0
                         [1.0, 0.26, 0.34, 0.49, 0.64]])    # same cell as row 0: the later row wins
    out = O.format_labels_tensor(rows, Sx, Sy)
    assert out.shape == (6, Sy, Sx)
    assert out[0].sum() == 2 and out[0, 1, 1] == 1 and out[0, 0, 3] == 1
    assert torch.equal(out[:, 1, 1], torch.tensor([1.0, 0.26, 0.34, 0.49, 0.64, 1.0]))
    assert torch.equal(out[:, 0, 3], torch.tensor([1.0, 0.80, 0.05, 0.95, 0.25, 5.0]))
    # a centre slightly left of the image wraps to the last column (Python indexing), exactly on the right edge it raises
    wrap = O.format_labels_tensor(torch.tensor([[0.0, -0.2, 0.4, 0.1, 0.6]]), Sx, Sy)
    assert wrap[0, 1, Sx - 1] == 1
    with pytest.raises(IndexError):
        O.format_labels_tensor(torch.tensor([[0.0, 0.9, 0.4, 1.1, 0.6]]), Sx, Sy)
    # cxcywh rows go through torchvision's box_convert first (label_file_to_tensor)
    t = O.label_rows_to_tensor(torch.tensor([[3.0, 0.375, 0.5, 0.15, 0.2]]), Sx, Sy)
    assert torch.equal(t[:, 1, 1], torch.tensor([1.0, 0.375 - 0.5 * 0.15, 0.5 - 0.5 * 0.2, 0.375 + 0.5 * 0.15, 0.5 + 0.5 * 0.2, 3.0]).float())
    assert torch.equal(O.label_rows_to_tensor(torch.zeros(0, 5), Sx, Sy), torch.zeros(6, Sy, Sx))


def test_oracle_flips_known_answers():
    img = torch.arange(2 * 1 * 2 * 3, dtype=torch.uint8).reshape(2, 1, 2, 3)
    lab = torch.zeros(2, 6, 2, 3)
    lab[0, :, 0, 2] = torch.tensor([1.0, 0.7, 0.1, 0.9, 0.3, 4.0])
    c = lambda v: float(1 - torch.tensor(v, dtype=torch.float32))   # 1 - x in fp32
    hi, hl = O.hflip_with_bbs(img, lab)
    assert torch.equal(hi[0, 0], torch.tensor([[2, 1, 0], [5, 4, 3]], dtype=torch.uint8))
    assert torch.equal(hl[0, :, 0, 0], torch.tensor([1.0, c(0.9), 0.1, c(0.7), 0.3, 4.0]))
    assert hl[0, 1, 1, 1] == 1.0 and hl[0, 3, 1, 1] == 1.0 and hl[0, 0, 1, 1] == 0.0   # empty cells: 1 - 0, mask untouched
    vi, vl = O.vflip_with_bbs(img, lab)
    assert torch.equal(vi[0, 0], torch.tensor([[3, 4, 5], [0, 1, 2]], dtype=torch.uint8))
    assert torch.equal(vl[0, :, 1, 2], torch.tensor([1.0, 0.7, c(0.3), 0.9, c(0.1), 4.0]))
    assert torch.equal(lab[0, :, 0, 2], torch.tensor([1.0, 0.7, 0.1, 0.9, 0.3, 4.0]))   # the oracle does not touch its input


def test_load_labels_and_label_indices(tmp_path):
    from yogo_amd import data as D

    classes = ["healthy", "ring", "troph"]
    p = tmp_path / "a.txt"
    p.write_text("0 0.5 0.5 0.05 0.05\n2 0.25 0.75 0.04 0.06\n1 0.1 0.1 0.001 0.001\n")   # last box is under the area filter
    assert D.load_labels(p, classes) == [[0.0, 0.5, 0.5, 0.05, 0.05], [2.0, 0.25, 0.75, 0.04, 0.06]]
    q = tmp_path / "b.csv"
    q.write_text("ring,0.5,0.5,0.05,0.05\ntroph,0.25,0.75,0.04,0.06\n")
    notes = {"categories": [{"id": 0, "name": "troph"}, {"id": 1, "name": "healthy"}]}
    assert [r[0] for r in D.load_labels(q, classes, notes)] == [1.0, 2.0]
    assert D.correct_label_idx("0", classes, notes) == 2 and D.correct_label_idx("1", classes, notes) == 0
    assert D.correct_label_idx("2", classes) == 2
    with pytest.raises(ValueError, match="not found in notes.json"):
        D.correct_label_idx("7", classes, notes)
    e = tmp_path / "empty.txt"
    e.write_text("")
    assert D.load_labels(e, classes) == []
    bad = tmp_path / "bad.txt"
    bad.write_text("0 0.5 0.5 0.05\n0 0.5 0.5 0.05\n")
    with pytest.raises(AssertionError, match="should have"):
        D.load_labels(bad, classes)
    assert D.AREA_FILTER_THRESHOLD == 200 / (772 * 1032) and D.LABEL_TENSOR_PRED_DIM_SIZE == 6


def test_data_step_refuses_cpu_tensors():
    from yogo_amd import data as D

    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        D.flip_batch(torch.zeros(1, 1, 4, 4, dtype=torch.uint8), torch.zeros(1, 6, 2, 2), True, False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        D.format_labels_batch([torch.zeros(1, 5)], 4, 4, device="cpu")


# ---------------------------------------------------------------------------------------------------------------------------
# GPU: HIP kernels against the oracle, bit for bit
# ---------------------------------------------------------------------------------------------------------------------------
def _random_rows(g, n, num_classes=7, cxcywh=False):
    c = torch.rand(n, 2, generator=g) * 0.96 + 0.02
    w = 0.0425 * torch.exp(torch.randn(n, generator=g) * 0.2)
    h = 0.0555 * torch.exp(torch.randn(n, generator=g) * 0.2)
    cls = torch.randint(0, num_classes, (n,), generator=g).float()
    if cxcywh:
        return torch.stack((cls, c[:, 0], c[:, 1], w, h), dim=1)
    return torch.stack((cls, c[:, 0] - w / 2, c[:, 1] - h / 2, c[:, 0] + w / 2, c[:, 1] + h / 2), dim=1)


@pytest.mark.gpu
@pytest.mark.parametrize("Sx,Sy", [(129, 97), (33, 25), (5, 3)])
def test_rasteriser_matches_oracle(Sx, Sy):
    from yogo_amd import data as D

    g = torch.Generator().manual_seed(Sx)
    counts = [64, 0, 300, 1, 17, 2000]          # 2000 rows on a coarse grid: many rows per cell, the last one must win
    rows = [_random_rows(g, n) for n in counts]
    got = D.format_labels_batch(rows, Sx, Sy)
    assert got.shape == (len(counts), 6, Sy, Sx) and got.is_cuda
    for b, r in enumerate(rows):
        assert torch.equal(got[b].cpu(), O.format_labels_tensor(r, Sx, Sy)), b
    rows_c = [_random_rows(g, n, cxcywh=True) for n in counts]
    got = D.format_labels_batch([r.cuda() for r in rows_c], Sx, Sy, "cxcywh")
    for b, r in enumerate(rows_c):
        assert torch.equal(got[b].cpu(), O.label_rows_to_tensor(r, Sx, Sy)), b
    # single-image entry point, negative wrap, IndexError
    one = _random_rows(g, 40)
    assert torch.equal(D.format_labels_tensor(one.cuda(), Sx, Sy).cpu(), O.format_labels_tensor(one, Sx, Sy))
    wrap = torch.tensor([[0.0, -0.2 / Sx, 0.4, 0.1 / Sx, 0.6]])
    assert torch.equal(D.format_labels_tensor(wrap.cuda(), Sx, Sy).cpu(), O.format_labels_tensor(wrap, Sx, Sy))
    with pytest.raises(IndexError):
        D.format_labels_tensor(torch.tensor([[0.0, 0.9, 0.4, 1.1, 0.6]]).cuda(), Sx, Sy)
    with pytest.raises(IndexError):
        D.format_labels_tensor(torch.tensor([[0.0, float("nan"), 0.4, 1.1, 0.6]]).cuda(), Sx, Sy)
    assert torch.equal(D.format_labels_batch([], Sx, Sy), torch.zeros(0, 6, Sy, Sx, device="cuda"))


@pytest.mark.gpu
def test_label_files_to_batch(tmp_path):
    from yogo_amd import data as D

    classes = [str(i) for i in range(7)]
    g = torch.Generator().manual_seed(5)
    paths, want = [], []
    for k, n in enumerate((12, 0, 50)):
        rows = _random_rows(g, n, cxcywh=True)
        p = tmp_path / f"{k}.txt"
        p.write_text("".join(f"{int(r[0])} {r[1]:.6f} {r[2]:.6f} {r[3]:.6f} {r[4]:.6f}\n" for r in rows.tolist()))
        paths.append(p)
        want.append(O.label_rows_to_tensor(torch.tensor(D.load_labels(p, classes)).reshape(-1, 5), 129, 97))
    got = D.label_files_to_batch(paths, 129, 97, classes)
    assert torch.equal(got.cpu(), torch.stack(want))
    assert torch.equal(D.label_file_to_tensor(paths[2], 129, 97, classes).cpu(), want[2])


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype", [((3, 1, 772, 1032), torch.uint8), ((2, 3, 37, 53), torch.uint8), ((2, 1, 40, 56), torch.float32),
                                         ((2, 1, 9, 7), torch.float32)])
def test_flips_match_oracle(shape, dtype):
    from yogo_amd import data as D

    g = torch.Generator().manual_seed(shape[2])
    B = shape[0]
    img = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8) if dtype == torch.uint8 else torch.randn(shape, generator=g)
    Sy, Sx = (97, 129) if shape[2] == 772 else (5, 7)
    lab = torch.stack([O.format_labels_tensor(_random_rows(g, 20), Sx, Sy) for _ in range(B)])
    for h, v in ((True, False), (False, True), (True, True), (False, False)):
        wi, wl = img, lab
        if h:
            wi, wl = O.hflip_with_bbs(wi, wl)
        if v:
            wi, wl = O.vflip_with_bbs(wi, wl)
        gi, gl = D.flip_batch(img.cuda(), lab.cuda(), h, v)
        assert torch.equal(gi.cpu(), wi) and torch.equal(gl.cpu(), wl), (h, v)
    # the reference's modules, one draw per transform and batch, fused into one pass by MultiArgSequential
    aug = D.MultiArgSequential(D.RandomHorizontalFlipWithBBs(0.5), D.DualInputId(), D.RandomVerticalFlipWithBBs(0.5))
    assert len(aug) == 2
    for seed in range(6):
        torch.manual_seed(seed)
        wi, wl = O.random_flips_with_bbs(img, lab)
        torch.manual_seed(seed)
        gi, gl = aug(img.cuda(), lab.cuda())
        assert torch.equal(gi.cpu(), wi) and torch.equal(gl.cpu(), wl), seed
        torch.manual_seed(seed)
        gi, gl = D.RandomVerticalFlipWithBBs(0.5)(*D.RandomHorizontalFlipWithBBs(0.5)(img.cuda(), lab.cuda()))
        assert torch.equal(gi.cpu(), wi) and torch.equal(gl.cpu(), wl), seed
    # collate: None items are dropped, the batch lands on the device
    items = [(img[b], lab[b]) for b in range(B)] + [None]
    torch.manual_seed(1)
    ci, cl = D.collate_batch_robust(items, aug, device="cuda")
    torch.manual_seed(1)
    wi, wl = O.random_flips_with_bbs(img, lab)
    assert torch.equal(ci.cpu(), wi) and torch.equal(cl.cpu(), wl)


# ---------------------------------------------------------------------------------------------------------------------------
# Golden vectors written by the REAL reference functions (tests/golden/make_golden_data.py imports yogo/data/yogo_dataset.py and
# yogo/data/data_transforms.py on stubs of their torchvision / ruamel imports): the oracle, the host-side parser and the HIP
# kernels are all held to them bit for bit.
# ---------------------------------------------------------------------------------------------------------------------------
def _golden():
    import json

    from _util import load_npz

    z = load_npz("data_step.npz")
    return z, json.loads(str(z["meta"]))


def test_oracle_rasteriser_matches_reference_vectors():
    z, meta = _golden()
    for c in meta["rast"]:
        rows = torch.from_numpy(z[f"rast/{c['name']}/rows"])
        want = torch.from_numpy(z[f"rast/{c['name']}/out"])
        assert torch.equal(O.format_labels_tensor(rows, c["Sx"], c["Sy"]), want), c
    for c in meta["rast_err"]:
        rows = torch.from_numpy(z[f"rast_err/{c['name']}/rows"])
        if c["raises"] == "IndexError":
            with pytest.raises(IndexError):
                O.format_labels_tensor(rows, c["Sx"], c["Sy"])
        else:
            O.format_labels_tensor(rows, c["Sx"], c["Sy"])


def test_label_files_match_reference_vectors(tmp_path):
    """the host-side parser (yogo_amd.data.load_labels: csv sniffing, header row, area filter, empty files) against the rows the
    reference's load_labels returned, and the oracle's rasteriser on those rows against the reference's label_file_to_tensor"""
    from yogo_amd import data as D

    z, meta = _golden()
    for f in meta["files"]:
        path = tmp_path / f["name"]
        path.write_text(f["text"])
        rows = D.load_labels(path, meta["classes"])
        want_rows = z[f"file/{f['name']}/rows"]
        assert np.array_equal(np.asarray(rows, dtype=np.float64).reshape(-1, 5), want_rows), f["name"]
        for (Sx, Sy) in ((129, 97), (33, 25)):
            want = torch.from_numpy(z[f"file/{f['name']}/{Sx}x{Sy}"])
            got = O.label_rows_to_tensor(torch.tensor(rows, dtype=torch.float32).reshape(-1, 5), Sx, Sy)
            assert torch.equal(got, want), (f["name"], Sx, Sy)


def test_oracle_flips_match_reference_vectors():
    z, meta = _golden()
    for c in meta["flips"]:
        n = c["name"]
        img, lab = torch.from_numpy(z[f"flip/{n}/img"]), torch.from_numpy(z[f"flip/{n}/lab"])
        hi, hl = O.hflip_with_bbs(img, lab)
        vi, vl = O.vflip_with_bbs(img, lab)
        bi, bl = O.vflip_with_bbs(*O.hflip_with_bbs(img, lab))
        for tag, (gi, gl) in (("h", (hi, hl)), ("v", (vi, vl)), ("hv", (bi, bl)), ("h0", (img, lab)), ("v0", (img, lab))):
            assert torch.equal(gi, torch.from_numpy(z[f"flip/{n}/{tag}/img"])), (n, tag)
            assert torch.equal(gl, torch.from_numpy(z[f"flip/{n}/{tag}/lab"])), (n, tag)


@pytest.mark.gpu
def test_hip_data_step_matches_reference_vectors(tmp_path):
    """labels_rasterize_kernel / flip kernels (data_aug.hip) against the vectors of the reference's own functions, bit for bit"""
    from yogo_amd import data as D

    z, meta = _golden()
    for c in meta["rast"]:
        rows = torch.from_numpy(z[f"rast/{c['name']}/rows"])
        want = torch.from_numpy(z[f"rast/{c['name']}/out"])
        assert torch.equal(D.format_labels_tensor(rows.cuda(), c["Sx"], c["Sy"]).cpu(), want), c
    for c in meta["rast_err"]:
        rows = torch.from_numpy(z[f"rast_err/{c['name']}/rows"])
        if c["raises"] == "IndexError":
            with pytest.raises(IndexError):
                D.format_labels_tensor(rows.cuda(), c["Sx"], c["Sy"])
    for f in meta["files"]:
        path = tmp_path / f["name"]
        path.write_text(f["text"])
        for (Sx, Sy) in ((129, 97), (33, 25)):
            got = D.label_file_to_tensor(path, Sx, Sy, meta["classes"])
            assert torch.equal(got.cpu(), torch.from_numpy(z[f"file/{f['name']}/{Sx}x{Sy}"])), (f["name"], Sx, Sy)
    for c in meta["flips"]:
        n = c["name"]
        img, lab = torch.from_numpy(z[f"flip/{n}/img"]), torch.from_numpy(z[f"flip/{n}/lab"])
        for tag, (h, v) in (("h", (True, False)), ("v", (False, True)), ("hv", (True, True))):
            gi, gl = D.flip_batch(img.cuda(), lab.cuda(), h, v)
            assert torch.equal(gi.cpu(), torch.from_numpy(z[f"flip/{n}/{tag}/img"])), (n, tag)
            assert torch.equal(gl.cpu(), torch.from_numpy(z[f"flip/{n}/{tag}/lab"])), (n, tag)
