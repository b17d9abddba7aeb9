"""GPU: gate-pinned gradient comparison.

The HIP training step's gradients sit up to a few percent (max-norm) from an fp64 evaluation of the same step, which the
round-2 tests accepted with a 6 % bar on the argument that ReLU / max-pool gates on noise-level pre-activations flip
between evaluation orders.  This test SHOWS it: the decisions the HIP forward actually took (sign of scale*z+shift per
layer, the winning row of every max-pool group; taken from the tensors its backward keeps, mlp.record_gates) are
pinned into the fp64 oracle network, which then evaluates exactly the piecewise-linear branch the HIP step
differentiated.  What remains is rounding: every parameter tensor must agree within 2e-3 of its max-norm, at the
parity size (B = 2) and at the benchmark size (B = 16).  A tensor that still stuck out would be a bug, not a flip."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SA = ("sa1", "sa2", "sa3", "sa4")
FP = ("fp4", "fp3", "fp2", "fp1")


def _gates_from_taps(torch, taps, B):
    """HIP rows are channel-last ([B*S*K, C] grouped rows, [B*N, C] point rows); the oracle is channel-first."""
    assert len(taps) == 8, len(taps)
    gates = {}
    for name, tap in zip(SA + FP, taps):
        masks = []
        for z, (scale, shift, _, _) in zip(tap["zs"], tap["coefs"]):
            on = (z * scale + shift) > 0                                   # what max(scale*z+shift, 0) let through
            masks.append(on)
        if name in SA:
            K = tap["pool_k"]
            S = masks[0].shape[0] // (B * K)
            relu = [m.reshape(B, S, K, -1).permute(0, 3, 2, 1).cpu() for m in masks]     # -> [B, C, K, S]
            pool = tap["argk"].reshape(B, S, -1).permute(0, 2, 1).unsqueeze(2).to(torch.int64).cpu()   # [B, C, 1, S]
            # the pooled output is relu(bn(z)) of the winner: its gate is the last layer's mask at the winner
            gates[name] = {"relu": relu, "pool": pool}
        else:
            N = masks[0].shape[0] // B
            relu = [m.reshape(B, N, -1).permute(0, 2, 1).cpu() for m in masks]           # -> [B, C, N]
            if name == "fp1" and len(relu) == 4:                                          # conv1 / bn1 ran as fp1's last layer
                gates["head"] = {"relu": [relu.pop()]}
            gates[name] = {"relu": relu}
    return gates


@pytest.mark.parametrize("B,kind", ((2, "cube"), (2, "facade"), (16, "cube")))
def test_gradients_with_pinned_gates_agree_to_rounding(orc, synth, B, kind):
    import torch
    from khairil_tum_facade_semantic_segmentation_amd import mlp, ops
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_utils as U
    K, C = 18, 9
    blocks, labels, starts, cw = synth.draw_case(41 if B == 2 else synth.BENCH_SEED, B, 4096, C, kind, K)
    filled = synth.fill_state_dict(orc.state_shapes(K, C - 6))
    model = M.get_model(K, C - 6)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model = model.cuda().train()
    model.drop1.p = 0.0
    with mlp.record_gates() as taps, U.fps_starts(starts):
        pred, tf = model(torch.from_numpy(blocks).cuda().permute(0, 2, 1))
    loss = M.get_loss()(pred.contiguous().view(-1, K), torch.from_numpy(labels).cuda().view(-1), tf, torch.from_numpy(cw).cuda())
    loss.backward()
    ops.check_errors()
    grads = {k: p.grad.cpu().numpy().astype(np.float64) for k, p in model.named_parameters()}
    gates = _gates_from_taps(torch, taps, B)
    del taps

    net = orc.OracleNet(filled, dropout_p=0.0, dtype=torch.float64)
    net.training = True
    net.gates = gates
    logp, _ = net.forward(blocks.transpose(0, 2, 1), starts)
    oloss = net.loss(logp, labels, cw)
    oloss.backward()
    assert abs(float(loss) - float(oloss.detach())) <= 1e-4
    worst = {}
    for k, p in net.named_parameters():
        ref = p.grad.numpy()
        if k.endswith(".bias") and ("mlp_convs" in k or k == "conv1.bias"):
            continue                                     # a conv bias under train-mode BatchNorm: exact gradient 0
        worst[k] = np.abs(grads[k] - ref).max() / (np.abs(ref).max() + 1e-30)
    bad = {k: v for k, v in worst.items() if v > 2e-3}
    assert not bad, "gradients differ beyond rounding although every gate is pinned: %s" % sorted(bad.items(), key=lambda kv: -kv[1])[:5]
    # and the same comparison WITHOUT pinning is the few-percent band the un-masked tests allow: the gap is the gates
    print("worst pinned max-norm distance %.2e (%s)" % (max(worst.values()), max(worst, key=worst.get)))
