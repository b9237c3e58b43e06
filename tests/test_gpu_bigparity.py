"""GPU: parity of the block-sharded prediction AT THE FULL SIZES of BASELINE.json configs 4 and 5 (bench.py's `sharded_2d` /
`sharded_3d` inputs) -- reference semantics: big == whole (tests/test_big.py:98-147, stardist/big.py:89-122).

  config 4, 16384^2:  (a) the compiled reference NMS over the union of the gathered survivors gives exactly the keep mask the
                      interior / band rule + band-restricted NMS produced; (b) the final instances and the label image equal the
                      committed golden of the reference NMS over ALL 32 M candidates of the whole slide
                      (tests/golden/sharded_fullsize.json, made by tests/golden/make_sharded_golden.py on the GPU box).
  config 5, 1024^3:   the compiled Qhull reference is out of reach at 10^7 candidates, so sharded == the monolithic HIP NMS over all
                      candidates of the whole volume (the kernel itself is pinned to the reference at 256^3 and below); and at 512^3 /
                      8 blocks the sharded result equals the committed golden of the compiled reference over all 1.3 M candidates."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden(key):
    path = os.path.join(ROOT, "tests", "golden", "sharded_fullsize.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/sharded_fullsize.json not generated yet")
    g = json.load(open(path))
    if key not in g:
        pytest.skip("no '%s' entry in sharded_fullsize.json" % key)
    return g[key]


def test_sharded_16384_equals_reference(refmods):
    import torch
    import _bigparity as B
    dev = torch.device("cuda:0")
    cfg = B.CFG2D
    model, big, axes = B.model_and_input(2, cfg, dev)
    labels, res = model.predict_instances_sharded(big, axes, block_size=cfg["block"], min_overlap=cfg["overlap"], context=cfg["context"], keep_debug=True)
    st, dbg = model._last_sharded_stats, model._last_sharded_debug
    assert st["blocks"] == 16 and st["unique"] > 500000 and st["interior"] > 0 and st["band"] > 0
    # (a) the reference over the union of the gathered survivors, in the final score order
    so = dbg["order"]
    d = dbg["dist"][so].cpu().numpy().astype(np.float32); p = dbg["points"][so].cpu().numpy().astype(np.float32)
    refmods.stardist2d(); refmods.set_threads(min(os.cpu_count() or 1, 32))
    ref_keep = refmods.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(model.thresholds.nms)).astype(bool)
    got = dbg["keep"][so].cpu().numpy()
    assert np.array_equal(ref_keep, got), "%d of %d flags differ (first: %s)" % ((ref_keep != got).sum(), len(got), np.flatnonzero(ref_keep != got)[:10])
    assert int(ref_keep.sum()) == len(res["prob"])
    # (b) the reference over ALL candidates of the whole slide
    g = _golden("2d")
    assert {k: g[k] for k in cfg} == cfg, "the golden was made for another block geometry"
    assert st["instances"] == g["survivors"], (st["instances"], g["survivors"])
    assert B.points_digest(res["points"]) == g["points_sha256"]
    assert B.array_digest(np.asarray(labels).astype(np.int32)) == g["labels_sha256"]


@pytest.mark.parametrize("cfgname", ["1024", "512-golden"])
def test_sharded_3d_equals_monolithic_and_reference(cfgname):
    import torch
    import _bigparity as B
    from stardist_amd.lib.stardist3d import c_non_max_suppression_inds
    from stardist_amd.rays3d import rays_from_json
    dev = torch.device("cuda:0")
    cfg = B.CFG3D if cfgname == "1024" else B.CFG3D_REF
    model, big, axes = B.model_and_input(3, cfg, dev)
    tiles, res = model.predict_instances_sharded(big, axes, block_size=cfg["block"], min_overlap=cfg["overlap"], context=cfg["context"], labels_out="local")
    st = dict(model._last_sharded_stats)
    assert st["blocks"] == 8 and st["instances"] > 1000 * (cfg["size"] // 256) ** 3
    del tiles
    torch.cuda.empty_cache()
    # the monolithic HIP NMS over every candidate of the whole volume, in predict_instances' order
    dist, prob, pts, nb = B.whole_input_candidates(model, big, axes, cfg)
    rays = rays_from_json(model.config.rays_json)
    verts = torch.as_tensor(np.ascontiguousarray(rays.vertices, np.float32), device=dev)
    faces = torch.as_tensor(np.ascontiguousarray(rays.faces, np.int32), device=dev)
    keep = c_non_max_suppression_inds(dist.float().contiguous(), pts.float().contiguous(), verts, faces, prob.float().contiguous(), 1, 1, 0,
                                      np.float32(model.thresholds.nms))
    keep = keep.cpu().numpy().astype(bool) if torch.is_tensor(keep) else np.asarray(keep, bool)
    mono = pts.cpu().numpy()[keep]
    print("3D %s: %d candidates, sharded %d instances, monolithic %d" % (cfgname, int(prob.numel()), st["instances"], int(keep.sum())))
    assert int(keep.sum()) == st["instances"]
    assert B.points_digest(mono) == B.points_digest(res["points"])
    if cfgname != "1024":
        g = _golden("3d")
        assert {k: g[k] for k in cfg} == cfg
        assert g["candidates"] == int(prob.numel()) and g["survivors"] == st["instances"]
        assert B.points_digest(res["points"]) == g["points_sha256"]
        assert B.array_digest(np.packbits(keep)) == g["keep_sha256"]
