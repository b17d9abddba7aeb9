#!/usr/bin/env python3
"""BASELINE configs[2] end to end on the HIP path: TUM-Facade LAS files -> trained pointnet2_sem_seg -> whole-scene labels
and per-class IoU of the test area.  What the reference does with sem_seg_training.py + sem_seg_testing.py (which stay as
they are but need laspy / open3d / h5py, absent here), restated on the package's pieces:

    las.read_las -> merge_labels_to_8 (--class8)                         sem_seg_training.py:137-169
    sample slots by point share, 70 / 30 split of the slots              :184-193, :434-441
    label_weights over all rooms                                         :264-278, :533-536
    MultiRoomSampler (device) -> train_epoch (rotate-z, captured step)   :200-259, localfunctions.py:184-227
    eval_epoch per epoch, BestModel (best_model.pth), model.pth / 5      localfunctions.py:229-322
    DeviceSceneTiler -> infer_scene (votes on the device) -> IoU, labels sem_seg_testing.py:182-254, localfunctions.py:349-479

    python tools/run_facade.py --data DIR --test-area NAME.las --epochs 25 [--class8] [--no-color] [--gpus N] [--oracle]
    python tools/run_facade.py --data DIR --test-area NAME.las --test-only [--checkpoint best_model.pth] ...   (sem_seg_testing.py alone)

--gpus N starts N ranks (one per GPU) under torch.distributed.run: every rank draws its own blocks (the rank is in the
sampler seed), ONE all-reduce of the packed gradients per step, and the test scene's sub-batches are sharded over the ranks
with one vote all-reduce per scene.  --oracle also runs the CPU oracle network (oracle/: test infrastructure) on the same
tiles with the same weights and prints the mIoU difference -- the comparator for the north_star's "mIoU within +-0.2 points".
Note the reference's inverted flag: its --RGB_OFF switch turns colour ON (sem_seg_training.py:352-355); here --no-color
means what it says."""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

CLASSES_18 = ["total", "wall", "window", "door", "balcony", "molding", "deco", "column", "arch", "drainpipe", "stairs",
              "ground surface", "terrain", "roof", "blinds", "outer ceiling surface", "interior", "other"]
CLASSES_8 = ["wall", "window", "door", "molding", "other", "terrain", "column", "arch"]      # sem_seg_training.py:48-50


def load_room(path, class8, color):
    """-> {"name", "xyz" [P,3] float64, "labels" [P] int64, "extra": [per-feature arrays], "feature_name": [...]}
    (the fields TrainCustomDataset / TestCustomDataset keep per room)."""
    from khairil_tum_facade_semantic_segmentation_amd import las
    d = las.read_las(path)
    labels = np.asarray(d.classification, dtype=np.int64)
    if class8:
        labels = las.merge_labels_to_8(labels)
        if (labels < 0).any():
            raise ValueError("%s: %d points carry classes outside the reference's 18 -> 8 mapping" % (path, int((labels < 0).sum())))
    extra, names = [], []
    if color:
        if d.red is None:
            raise ValueError("%s has no colour (point format %d): run with --no-color" % (path, d.header["point_format"]))
        extra = [np.asarray(d.red, dtype=np.float64), np.asarray(d.blue, dtype=np.float64), np.asarray(d.green, dtype=np.float64)]
        names = ["red", "blue", "green"]                       # the reference's order, sem_seg_training.py:123-126
    return {"name": os.path.basename(path), "xyz": d.xyz(), "labels": labels, "extra": extra, "feature_name": names}


def sample_slots(points_per_room, num_point, sample_rate=1.0):
    """room_idxs of TrainCustomDataset.__init__ (sem_seg_training.py:184-193): one slot per num_point points, every
    room's index repeated by its share of the points."""
    n = np.asarray(points_per_room, dtype=np.float64)
    prob = n / n.sum()
    num_iter = int(n.sum() * sample_rate / num_point)
    idxs = []
    for r in range(len(n)):
        idxs.extend([r] * int(round(prob[r] * num_iter)))
    return np.asarray(idxs, dtype=np.int64)


def split_slots(room_idxs, num_rooms, train_ratio=0.7, seed=0):
    """random_split(range(len(dataset)), [train, eval]) of sem_seg_training.py:434-441 -- both halves keep drawing random
    blocks from ALL rooms; what the split decides is how many slots of each room a half holds.
    -> (train slots per room [R], eval slots per room [R])"""
    import torch
    g = torch.Generator().manual_seed(int(seed))
    perm = torch.randperm(len(room_idxs), generator=g).numpy()
    ntrain = int(train_ratio * len(room_idxs))
    tr = np.bincount(room_idxs[perm[:ntrain]], minlength=num_rooms)
    ev = np.bincount(room_idxs[perm[ntrain:]], minlength=num_rooms)
    return tr, ev


def init_weights(model):
    """weights_init of sem_seg_training.py:554-561: xavier-normal Conv2d / Linear weights, zero biases (Conv1d layers
    keep torch's default initialisation there too: the class-name test only matches 'Conv2d' and 'Linear')."""
    import torch
    for m in model.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
            torch.nn.init.xavier_normal_(m.weight.data)
            torch.nn.init.constant_(m.bias.data, 0.0)


def oracle_scene_labels(state, data, index, weight, num_points, num_classes, batch_size):
    """The CPU oracle network (oracle/pn2_oracle.OracleNet: the reference's forward restated on torch CPU + C index ops) and
    the reference's add_vote on the given tiles -- test infrastructure, reached through --oracle only.  FPS start indices
    per sub-batch from RandomState(0), the sequence run() injects into the HIP side."""
    import torch
    from oracle import pn2_oracle as orc
    from oracle import scene_oracle
    orc.build()
    net = orc.OracleNet({k: v.detach().cpu().numpy() for k, v in state.items()}, dropout_p=0.0)
    net.training = False
    pool = np.zeros((num_points, num_classes))
    nb = data.shape[0]
    rs = np.random.RandomState(0)
    with torch.no_grad():
        for s in range(0, nb, batch_size):
            blk = np.ascontiguousarray(data[s:min(s + batch_size, nb)].transpose(0, 2, 1)).astype(np.float32)
            starts = [rs.randint(0, n, size=(blk.shape[0],)) for n in (blk.shape[2], 1024, 256, 64)]
            logp, _ = net.forward(blk, starts)
            pred = logp.argmax(-1).numpy()
            e = s + blk.shape[0]
            scene_oracle.add_vote(pool, index[s:e], pred, weight[s:e])
    return np.argmax(pool, 1)


def spawn_ranks(args):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def run(args, log=print):
    import torch
    import torch.distributed as dist
    from khairil_tum_facade_semantic_segmentation_amd import ops, scene, train
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(args.backend, rank=rank, world_size=world)     # lazy communicator: created after the capture
    say = log if rank == 0 else (lambda *a, **k: None)
    classes = CLASSES_8 if args.class8 else CLASSES_18
    C = len(classes)
    color = not args.no_color

    files = sorted(glob.glob(os.path.join(args.data, "*.las")))
    train_files = [f for f in files if not f.endswith(args.test_area)]                     # sem_seg_training.py:359
    test_files = [f for f in files if f.endswith(args.test_area)]
    if not test_files or not (train_files or args.test_only):
        raise SystemExit("need at least one training .las and the test area %r under %s (found %d files)"
                         % (args.test_area, args.data, len(files)))
    os.makedirs(args.out, exist_ok=True)
    best_path = os.path.join(args.out, "best_model.pth")
    history = []
    if args.test_only:
        # sem_seg_testing.py: no training; the checkpoint the training driver left (its 'model_state_dict', :496-497)
        E = 3 if color else 0
        model = M.get_model(C, E).to(dev)
        ck = train.load_checkpoint(args.checkpoint or best_path, model)
        say("loaded %s (epoch %s, class_avg_iou %s)" % (args.checkpoint or best_path, ck.get("epoch"), ck.get("class_avg_iou")))
        engine = scene.BlockInferencer(model, args.batch_size, 6 + E, args.npoint)
    else:
        t0 = time.time()
        rooms = [load_room(f, args.class8, color) for f in train_files]
        say("read %d training rooms, %d points, in %.1f s" % (len(rooms), sum(r["xyz"].shape[0] for r in rooms), time.time() - t0))
        E = len(rooms[0]["feature_name"])
        weights = train.label_weights([r["labels"] for r in rooms], C, device=dev)              # :264-278, :533-536
        slots = sample_slots([r["xyz"].shape[0] for r in rooms], args.npoint)
        tr_slots, ev_slots = split_slots(slots, len(rooms), 0.7, args.seed)
        steps = int(tr_slots.sum()) // args.batch_size if args.steps_per_epoch is None else args.steps_per_epoch   # drop_last=True, :524-528
        eval_steps = int(ev_slots.sum()) // args.batch_size if args.eval_steps is None else args.eval_steps
        if steps < 1:
            raise SystemExit("the training rooms hold fewer than one batch of %d x %d-point slots" % (args.batch_size, args.npoint))
        say("%d slots (%d train, %d eval): %d training steps per epoch per rank, %d evaluation batches; class weights %s"
            % (len(slots), int(tr_slots.sum()), int(ev_slots.sum()), steps, eval_steps, np.round(weights.cpu().numpy(), 3).tolist()))
        samplers = [scene.DeviceBlockSampler(r["xyz"], r["labels"], r["extra"], r["feature_name"], args.npoint, device=dev) for r in rooms]
        tr_sampler, ev_sampler = scene.MultiRoomSampler(samplers), scene.MultiRoomSampler(samplers)
        tr_sampler.sizes = [max(float(v), 1e-9) for v in tr_slots]        # a block's room is drawn by the split's share of slots
        ev_sampler.sizes = [max(float(v), 1e-9) for v in ev_slots]

        torch.manual_seed(args.seed)
        model = M.get_model(C, E)
        init_weights(model)
        model = model.to(dev)
        trainer = train.SemSegTrainer(model, lr=args.learning_rate, weight_decay=args.decay_rate, class_weight=weights, graphs=True,
                                      prefetch_geometry=True, augment=True, metrics=True)
        start_epoch = 0
        if args.resume and os.path.exists(best_path):
            start_epoch = int(train.load_checkpoint(best_path, model, trainer)["epoch"])      # sem_seg_training.py:566-570
            say("resumed from %s at epoch %d" % (best_path, start_epoch))
        if use_dist:
            x0, y0 = train.draw_batch(tr_sampler, args.batch_size, args.seed, 0, 0, rank)
            trainer.prepare(x0, y0)                                        # capture BEFORE the first collective
        trainer.broadcast_parameters()
        best = train.BestModel(best_path if rank == 0 else None)
        engine = scene.BlockInferencer(model, args.batch_size, 6 + E, args.npoint)
        for epoch in range(start_epoch, args.epochs):
            t0 = time.time()
            model.train()
            tr = train.train_epoch(trainer, tr_sampler, epoch, steps, args.batch_size, seed=args.seed, learning_rate=args.learning_rate,
                                   lr_decay=args.lr_decay, step_size=args.step_size)
            torch.cuda.synchronize(dev)
            t_train = time.time() - t0
            # an empty ball-query neighbourhood is an IndexError in the reference (pointnet2_utils.py:59); here it is counted on
            # the device and raised once per epoch (raw coordinates whose squares swamp r^2 in fp32 are the usual cause)
            ops.check_errors(dev, "training epoch %d" % epoch)
            if epoch % 5 == 0 and rank == 0:                               # localfunctions.py:229-239
                train.save_checkpoint(os.path.join(args.out, "model.pth"), epoch, model, trainer)
            ev = None
            if eval_steps > 0:
                # the evaluation half draws random blocks too (the reference's eval dataset is the same class); every rank
                # draws the same ones (no rank in the seed) from its own replica
                batches = [train.draw_batch(ev_sampler, args.batch_size, args.seed + 7919, epoch, i, 0) for i in range(eval_steps)]
                if train.gave_up_blocks(ev_sampler):
                    raise RuntimeError("the block sampler gave up on evaluation blocks: a room without a 1 m column of > 1024 points")
                ev = train.eval_epoch(model, batches, class_weight=weights, engine=engine)
                del batches
                best.update(epoch, ev["mIoU"], model, trainer)
            rec = {"epoch": epoch, "train_loss": tr["loss"], "train_accuracy": tr.get("accuracy"), "lr": tr["lr"],
                   "bn_momentum": tr["bn_momentum"], "train_seconds": t_train,
                   "train_points_per_s": world * steps * args.batch_size * args.npoint / t_train}
            if ev is not None:
                rec.update({"eval_loss": ev["loss"], "eval_mIoU": ev["mIoU"], "eval_accuracy": ev["accuracy"], "best_mIoU": best.best_iou})
            history.append(rec)
            say(json.dumps(rec))
        if best.state is not None:
            model.load_state_dict(best.state)                              # sem_seg_testing.py:496-497 loads best_model.pth
        trainer.broadcast_parameters()                                     # BatchNorm statistics are rank-local: rank 0's model votes

    # ---- whole-scene test (sem_seg_testing.py + modelTesting)
    results = {"history": history, "classes": classes, "world_size": world, "scenes": []}
    tot = {k: np.zeros(C) for k in ("class_seen", "class_correct", "class_union")}
    for f in test_files:
        room = load_room(f, args.class8, color)
        P = room["xyz"].shape[0]
        lw = np.histogram(room["labels"], range(C + 1))[0].astype(np.float32)
        lw = lw / lw.sum()
        with np.errstate(divide="ignore"):
            lw = np.power(np.amax(lw) / lw, 1 / 3.0)                   # sem_seg_testing.py:171-178
        tiler = scene.DeviceSceneTiler(room["xyz"], room["labels"], room["extra"], room["feature_name"], lw, args.npoint, device=dev)
        votes_seed = [args.seed]

        def tile():
            votes_seed[0] += 1
            return tiler.tile(votes_seed[0])
        t0 = time.time()
        data, _, wt, idx = tile()
        pred, pool = scene.infer_scene(model, data, idx, wt, P, C, batch_size=args.batch_size, num_votes=args.num_votes,
                                       retile=tile, return_votes=True, engine=engine)
        torch.cuda.synchronize(dev)
        dt = time.time() - t0
        ops.check_errors(dev, "whole-scene inference of %s" % room["name"])
        m = scene.scene_metrics(pred, room["labels"], C)
        for k in tot:
            tot[k] += m[k]
        name = room["name"][:-4]
        if rank == 0:
            np.savetxt(os.path.join(args.out, name + ".txt"), pred.cpu().numpy().astype(np.int64), fmt="%d")   # localfunctions.py:423-427
        srec = {"scene": name, "points": P, "blocks": int(data.shape[0]), "votes": args.num_votes, "seconds": dt,
                "scene_mIoU": m["scene_mIoU"], "IoU": m["IoU"].tolist(), "labels_written": int(pred.numel()),
                "vote_pool_total": int(pool.sum().item()),
                "label_checksum": int((pred.to(torch.int64) * (torch.arange(P, device=pred.device) % 8191 + 1)).sum().item())}
        if args.oracle:
            # HIP against the CPU oracle network on ONE tiling with the same weights and the same FPS start indices (both
            # sides draw them from RandomState(0) per sub-batch); every rank runs the HIP side (its vote all-reduce is a
            # collective), rank 0 the oracle
            from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_utils as U
            data1, _, wt1, idx1 = tiler.tile(args.seed + 1)
            nb = data1.shape[0] if args.oracle_max_blocks is None else min(data1.shape[0], args.oracle_max_blocks)
            rs = np.random.RandomState(0)
            starts = []
            for s0 in range(0, nb, args.batch_size):
                b = min(args.batch_size, nb - s0)
                starts += [rs.randint(0, n, size=(b,)) for n in (args.npoint, 1024, 256, 64)]
            if world > 1:
                raise SystemExit("--oracle compares one replica with the CPU network: run it with --gpus 1")
            with U.fps_starts(starts):
                hip1 = scene.infer_scene(model, data1[:nb], idx1[:nb], wt1[:nb], P, C, batch_size=args.batch_size)
            o = oracle_scene_labels(model.state_dict(), data1[:nb].cpu().numpy(), idx1[:nb].cpu().numpy(), wt1[:nb].cpu().numpy(),
                                    P, C, args.batch_size)
            voted = np.zeros(P, dtype=bool)
            voted[idx1[:nb].cpu().numpy().reshape(-1)] = True
            mo = scene.scene_metrics(o[voted], room["labels"][voted], C)
            mh = scene.scene_metrics(hip1.cpu().numpy()[voted], room["labels"][voted], C)
            srec.update({"oracle_blocks": int(nb), "oracle_scene_mIoU": mo["scene_mIoU"], "hip_scene_mIoU_same_tiles": mh["scene_mIoU"],
                         "hip_minus_oracle_scene_mIoU": mh["scene_mIoU"] - mo["scene_mIoU"],
                         "hip_minus_oracle_mIoU": mh["mIoU"] - mo["mIoU"],
                         "label_agreement": float((o[voted] == hip1.cpu().numpy()[voted]).mean())})
        results["scenes"].append(srec)
        say(json.dumps(srec))
    iou = tot["class_correct"] / (tot["class_union"] + 1e-6)
    results["IoU"] = iou.tolist()
    results["mIoU"] = float(iou.mean())                                # 'eval point avg class IoU', localfunctions.py:477
    results["avg_class_acc"] = float((tot["class_correct"] / (tot["class_seen"] + 1e-6)).mean())
    results["accuracy"] = float(tot["class_correct"].sum() / (tot["class_seen"].sum() + 1e-6))
    say("------- IoU --------")
    for name, v in zip(classes, iou):
        say("class %-22s IoU: %.3f" % (name, v))
    say("eval point avg class IoU: %f" % results["mIoU"])
    if rank == 0:
        with open(os.path.join(args.out, "results.json"), "w") as fh:
            json.dump(results, fh, indent=1)
    if use_dist and args.destroy_group:
        dist.destroy_process_group()
    return results


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data", required=True, help="directory with the .las files (the reference's --rootdir)")
    ap.add_argument("--test-area", default="DEBY_LOD2_4959323.las", help="file-name suffix of the held-out scene (sem_seg_training.py:359)")
    ap.add_argument("--out", default="log/facade", help="checkpoints, <scene>.txt label files, results.json")
    ap.add_argument("--epochs", type=int, default=25)
    ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--npoint", type=int, default=4096)
    ap.add_argument("--learning-rate", type=float, default=1e-3)
    ap.add_argument("--decay-rate", type=float, default=1e-4)
    ap.add_argument("--lr-decay", type=float, default=0.7)
    ap.add_argument("--step-size", type=int, default=10)
    ap.add_argument("--num-votes", type=int, default=1)
    ap.add_argument("--class8", action="store_true")
    ap.add_argument("--no-color", action="store_true", help="geometry-only 4096x6 blocks (BASELINE configs[3])")
    ap.add_argument("--steps-per-epoch", type=int, default=None, help="default: the 70 %% share of the sample slots / batch size")
    ap.add_argument("--eval-steps", type=int, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--test-only", action="store_true",
                    help="sem_seg_testing.py's job alone: load --checkpoint (default <out>/best_model.pth) and label the test area")
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--oracle", action="store_true", help="also run the CPU oracle network on the test scene's tiles (slow)")
    ap.add_argument("--oracle-max-blocks", type=int, default=None)
    ap.add_argument("--keep-group", dest="destroy_group", action="store_false")
    return ap.parse_args(argv)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    run(args)


if __name__ == "__main__":
    main()
