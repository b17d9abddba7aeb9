"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/pn2_hip.h declares; argument validation happens before any launch (no GPU
needed for the error paths); the Python surface mirrors the reference module's names."""
import ctypes
import inspect
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as entry
    entry.build()
    from khairil_tum_facade_semantic_segmentation_amd import _lib
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(REPO, "include", "pn2_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pn2_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from khairil_tum_facade_semantic_segmentation_amd import _lib
    names = declared_symbols()
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names          # the ctypes table covers the header, nothing else
    assert lib.pn2_abi_version() == 1


def test_argument_validation_without_gpu(lib):
    null = None
    one = ctypes.c_void_p(16)                        # never dereferenced: validation fails first
    assert lib.pn2_farthest_point_sample(null, 1, 8, 2, one, one, null, null, null) == -1
    assert lib.pn2_farthest_point_sample(one, 1, 0, 2, one, one, null, null, null) == -2
    assert lib.pn2_farthest_point_sample(one, 1, 40000, 2, one, one, null, null, null) == -3
    assert lib.pn2_farthest_point_sample(one, 0, 8, 2, one, one, null, null, null) == 0      # empty batch
    assert lib.pn2_ball_query_group(0.1, 65, one, one, null, 1, 8, 2, 0, one, null, 0, null, null) == -3
    assert lib.pn2_ball_query_group(0.1, 8, one, one, null, 1, 8, 2, 3, one, null, 0, null, null) == -1   # D>0, no points
    assert lib.pn2_ball_query_group(0.1, 0, one, one, null, 1, 8, 2, 0, one, null, 0, null, null) == -2
    assert lib.pn2_ball_query_group(0.1, 8, one, one, one, 1, 8, 2, 3, one, one, 5, null, null) == -2    # pitch < 3+D
    assert lib.pn2_three_nn(one, one, 1, 8, 2, one, null, one, null) == -2                   # S < 3
    assert lib.pn2_index_points_backward(one, one, 1, 8, 4, 3, 6, 3, one, null) == -2        # col0+D > Cg
    assert lib.pn2_index_points(one, one, 1, 8, 4, 0, one, null, null) == 0                  # M == 0
    assert b"NULL" in lib.pn2_error_string(-1)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from khairil_tum_facade_semantic_segmentation_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.Pn2LibraryError):
        _lib.load()


def test_cpu_tensor_is_refused_not_silently_computed():
    import torch
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_utils as U
    with pytest.raises(RuntimeError, match="no CPU"):
        U.query_ball_point(0.1, 4, torch.zeros(1, 8, 3), torch.zeros(1, 2, 3))


def test_python_surface_mirrors_reference_names():
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_utils as U
    # names + leading positional parameters of reference models/pointnet2_utils.py
    want = {
        "square_distance": ["src", "dst"],
        "index_points": ["points", "idx"],
        "farthest_point_sample": ["xyz", "npoint"],
        "query_ball_point": ["radius", "nsample", "xyz", "new_xyz"],
        "sample_and_group": ["npoint", "radius", "nsample", "xyz", "points", "returnfps"],
        "sample_and_group_all": ["xyz", "points"],
    }
    for name, params in want.items():
        got = list(inspect.signature(getattr(U, name)).parameters)
        assert got[:len(params)] == params, name
    assert list(inspect.signature(U.PointNetSetAbstraction.__init__).parameters)[1:] == [
        "npoint", "radius", "nsample", "in_channel", "mlp", "group_all"]
    assert list(inspect.signature(U.PointNetSetAbstractionMsg.__init__).parameters)[1:] == [
        "npoint", "radius_list", "nsample_list", "in_channel", "mlp_list"]
    assert list(inspect.signature(U.PointNetFeaturePropagation.__init__).parameters)[1:] == ["in_channel", "mlp"]
    assert list(inspect.signature(U.PointNetFeaturePropagation.forward).parameters)[1:] == [
        "xyz1", "xyz2", "points1", "points2"]
    assert list(inspect.signature(M.get_model.__init__).parameters)[1:] == ["num_classes", "num_extra_features"]
    assert list(inspect.signature(M.get_loss.forward).parameters)[1:] == ["pred", "target", "trans_feat", "weight"]


def test_state_dict_keys_match_reference_layout(orc):
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
    for K, extra in ((18, 3), (8, 0)):
        sd = M.get_model(K, extra).state_dict()
        shapes = orc.state_shapes(K, extra)            # asserted equal to the reference's in make_golden.py
        assert list(sd.keys()) == list(shapes.keys())
        assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
