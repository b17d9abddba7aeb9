"""The plain-PointNet control model (BASELINE configs[4]; reference models/pointnet_sem_seg.py, pointnet_utils.py):
same state_dict as the reference (CPU check against the key list stored with the golden) and, on the GPU, its
outputs against the reference-generated golden (oracle/make_golden_pointnet.py)."""
import numpy as np
import pytest


def _model(K, C):
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet_sem_seg as P
    return P, P.get_model(K, C - 6)


def test_state_dict_keys_match_the_reference(golden):
    g = golden("pointnet_control")
    B, N, C, K = (int(v) for v in g["shape"])
    _, model = _model(K, C)
    assert sorted(model.state_dict().keys()) == [str(k) for k in g["keys"]]
    from khairil_tum_facade_semantic_segmentation_amd.models.pointnet_sem_seg import macs_per_point
    assert abs(macs_per_point(9, 18) - 1.146e6) < 0.01e6            # ~1.146 MMAC per point


@pytest.mark.gpu
def test_control_model_matches_reference_golden(golden, synth):
    import torch
    g = golden("pointnet_control")
    B, N, C, K = (int(v) for v in g["shape"])
    blocks, labels, _, cw = synth.draw_case(int(g["seed"]), B, N, C, "cube", K)
    P, model = _model(K, C)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model = model.cuda()
    x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1))).cuda()
    model.eval()
    with torch.no_grad():
        logp, tf = model(x)
    assert np.abs(logp.cpu().numpy() - g["eval_logp"]).max() <= 1e-3
    assert np.abs(tf.cpu().numpy() - g["eval_trans_feat"]).max() <= 1e-3
    model.train()
    logp, tf = model(x)
    loss = P.get_loss()(logp.reshape(-1, K), torch.from_numpy(labels).cuda().view(-1), tf, torch.from_numpy(cw).cuda())
    loss.backward()
    assert abs(float(loss.detach()) - float(g["train_loss"])) <= 1e-3
    # train mode: bn4 / bn5 of the two T-nets normalise over the B = 16 block vectors only, which amplifies rounding
    # differences of the 1024-wide max-pooled features by orders of magnitude (eval mode above meets 1e-3)
    assert np.abs(logp.detach().cpu().numpy()[:, ::16] - g["train_logp_sample"]).max() <= 2e-2
    params = dict(model.named_parameters())
    for key in g:
        if key.startswith("grad:"):
            ref = g[key].astype(np.float64)
            got = params[key[5:]].grad.cpu().numpy().reshape(-1)[::7].astype(np.float64)
            # the band two fp32 evaluation orders sit in (tests/test_hip_parity.py::_assert_gradient_close)
            assert np.linalg.norm(got - ref) <= 5e-2 * np.linalg.norm(ref) + 1e-9, (key, np.linalg.norm(got - ref) / np.linalg.norm(ref))
            assert np.abs(got - ref).max() <= 1e-1 * np.abs(ref).max() + 1e-9, key
    sd = model.state_dict()
    for key in g:
        if key.startswith("buf:"):
            assert np.abs(sd[key[4:]].cpu().numpy() - g[key]).max() <= 1e-3 * (1.0 + np.abs(g[key]).max()), key
